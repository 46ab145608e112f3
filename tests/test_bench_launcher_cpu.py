"""bench.py --gpus N without an external launcher: the script spawns its own N ranks (torch.distributed.run, one per GPU)
and refuses -- instead of quietly timing one rank -- when the node shows fewer GPUs than ranks."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_gpus_2_without_gpus_refuses():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has two GPUs: the refusal does not apply")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "refusing to share devices" in out.stderr and "--gpus 2" in out.stderr
    assert out.stdout.strip() == ""          # no bench line: a one-rank run must not pass for a two-GPU record


def test_self_launch_spawns_one_rank_per_gpu(monkeypatch):
    m = _bench_module()
    seen = {}
    monkeypatch.setattr(m, "visible_gpus", lambda: 8)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    assert m.self_launch(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_self_launch_refuses_more_ranks_than_gpus(monkeypatch, capsys):
    m = _bench_module()
    monkeypatch.setattr(m, "visible_gpus", lambda: 1)
    monkeypatch.setattr(subprocess, "call", lambda *a, **k: pytest.fail("must not spawn"))
    assert m.self_launch(2) != 0
    assert "refusing" in capsys.readouterr().err


def test_bench_refuses_stray_alignnet_environment():
    """Round 3's library read ALIGNNET_DBG & co. from the environment on every launch (work skipped inside the kernels, one variable away
    from a benchmark).  The library reads none now; bench.py refuses to run when one is set, before it touches a GPU."""
    env = dict(os.environ, ALIGNNET_DBG="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 2 and "ALIGNNET_DBG" in out.stderr and out.stdout.strip() == ""


def test_pmc_traffic_per_step_is_per_launch_times_launches_per_step():
    """tools/summarize_prof.py: the PMC passes are separate runs whose time-based spin-up makes different numbers of steps; round 3 divided
    their bytes by the TRACE run's step count.  Per step = per launch x launches per step of the same pass."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import summarize_prof as S
    pmc = {"pointnet_fused": {"FETCH_SIZE": {"sum": 4540.0 * 381, "dispatches": 381}, "WRITE_SIZE": {"sum": 1170.0 * 396, "dispatches": 396}},
           "fc_mfma": {"FETCH_SIZE": {"sum": 100.0 * 1143, "dispatches": 1143}, "WRITE_SIZE": {"sum": 10.0 * 1188, "dispatches": 1188}}}
    t = S.traffic_summary(pmc, 127, 132, {"pairs_per_gpu": 256, "num_points": 1024}, "t", "c")
    k = t["kernels"]["pointnet_fused"]
    assert abs(k["launches_per_step"] - 3.0) < 1e-9
    assert abs(k["hbm_bytes_per_launch"] - (2 * 4540.0 + 1170.0) * 1024) < 1e-6
    assert abs(k["hbm_bytes_per_step"] - 3 * k["hbm_bytes_per_launch"]) < 1e-6
    assert abs(t["hbm_bytes_per_step"] - sum(v["hbm_bytes_per_step"] for v in t["kernels"].values())) < 1e-6
    # and the committed summaries obey it
    import glob
    import json
    for f in glob.glob(os.path.join(ROOT, "profiles", "r0[45]_*pmc_traffic.json")):
        for name, v in json.load(open(f))["kernels"].items():
            assert abs(v["hbm_bytes_per_step"] - v["hbm_bytes_per_launch"] * v["launches_per_step"]) <= 1e-6 * max(1.0, v["hbm_bytes_per_step"]), (f, name)


def test_steps_of_counts_every_step_the_profiled_process_ran(tmp_path):
    """bench.py ran the K timed steps a second time with the kernel timers off and steps_of() did not count them, so every
    per-step figure was inflated (the line said 3.0 launches per step, the summary 3.5).  The bench line now carries `steps_executed`;
    older lines are rebuilt from their parts including the repeat."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import json
    import summarize_prof as S
    new = {"steps": 20, "warmup": 3, "spinup_steps_untimed": 96, "steps_executed": 139, "config": {"pairs_per_gpu": 256, "num_points": 1024}}
    old = {"steps": 20, "warmup": 3, "spinup_steps_untimed": 96, "without_kernel_timers": {"value": 1.0}, "config": {"pairs_per_gpu": 256, "num_points": 1024}}
    r3 = {"steps": 20, "warmup": 3, "spinup_steps_untimed": 96, "config": {"pairs_per_gpu": 256, "num_points": 1024}}
    for j, want in ((new, 139), (old, 139), (r3, 119)):
        p = tmp_path / "b.log"
        p.write_text("noise\n" + json.dumps(j) + "\n")
        n, shape = S.steps_of(str(p))
        assert n == want and shape == {"pairs_per_gpu": 256, "num_points": 1024}


def test_committed_summaries_agree_with_their_bench_lines_on_launches_per_step():
    """Each committed PMC traffic summary of round 5 on: the dominant kernel's launches per step equals what the bench line of the same
    profiled command reports (3.0 for the three-stage backbones) to 1 % -- the check that would have caught rounds 3 and 4."""
    import glob
    import json
    import re
    checked = 0
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))):
        m = re.fullmatch(r"r(\d+)_(.*?)_?pmc_traffic\.json", os.path.basename(f))
        if int(m.group(1)) < 5:
            continue
        line_f = os.path.join(ROOT, "profiles", "r%s_%sbench_under_rocprof.json" % (m.group(1), m.group(2) + "_" if m.group(2) else ""))
        if not os.path.exists(line_f):
            continue
        roof = json.load(open(line_f))["roofline"]
        kern = json.load(open(f))["kernels"]
        hit = sum(v["launches_per_step"] for name, v in kern.items() if roof["kernel"] in name)   # (all instantiations of that kernel)
        assert hit > 0, (f, roof["kernel"])
        assert abs(hit - roof["launches_per_step"]) <= 0.01 * roof["launches_per_step"], (f, hit, roof["launches_per_step"])
        checked += 1
    print("summaries checked:", checked)


def test_thread_ranks_rendezvous_and_missing_rank():
    """bench.py --rehearse-world: the stand-in for torch.distributed among rank THREADS (max / sum all-reduce, all-gather, object broadcast), and its
    point -- a collective that one rank does not enter times out instead of completing."""
    import threading
    import torch
    m = _bench_module()
    W = 3
    shared = m.ThreadRanks.Shared(W, timeout=20.0)
    got = [None] * W

    def body(r):
        d = m.ThreadRanks(r, shared)
        t = torch.tensor([float(r + 1)], dtype=torch.float64)
        d.all_reduce(t, op=d.ReduceOp.MAX)
        s = torch.tensor([float(r + 1)])
        d.all_reduce(s)
        out = [torch.empty(1) for _ in range(W)]
        d.all_gather(out, torch.tensor([10.0 * r]))
        box = ["id-from-0" if r == 0 else None]
        d.broadcast_object_list(box, src=0)
        d.barrier()
        got[r] = (float(t), float(s), [float(o) for o in out], box[0], d.get_rank(), d.get_world_size())
    th = [threading.Thread(target=body, args=(r,)) for r in range(W)]
    [t.start() for t in th]; [t.join() for t in th]
    for r in range(W):
        assert got[r] == (3.0, 6.0, [0.0, 10.0, 20.0], "id-from-0", r, W), got[r]
    # rank 1 skips the collective: the others must fail, not wait for ever
    shared2 = m.ThreadRanks.Shared(2, timeout=1.0)
    failed = []

    def lonely():
        try:
            m.ThreadRanks(0, shared2).all_reduce(torch.zeros(1))
        except threading.BrokenBarrierError:
            failed.append(True)
    t = threading.Thread(target=lonely); t.start(); t.join(30)
    assert failed == [True]
