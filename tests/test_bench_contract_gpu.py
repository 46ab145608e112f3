"""The driver's contract for bench.py: one JSON line on stdout with the agreed keys, the roofline and cpu_baseline objects, and
internally consistent numbers (value = pairs / time).  Short runs -- this checks the line, not the speed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def _run(*args, **extra_env):
    env = dict(os.environ, **extra_env)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout, got %d" % len(lines)
    return json.loads(lines[0])


def _check_common(d, steps, warmup):
    for k in REQUIRED:
        assert k in d, k
    assert d["unit"] == "pairs/s" and d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    pairs = d["config"]["pairs_per_gpu"]
    assert abs(d["value"] - pairs / (d["ms_per_step"] * 1e-3)) <= 2e-3 * d["value"]          # value = pairs / time
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma", "valu") and r["unit"] == {"hbm": "GB/s", "mfma": "TFLOP/s", "valu": "Ginstr/s"}[r["bound"]]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3 and 0.0 < r["frac"] < 1.0
    # every call of the headline step is accounted for (tools/summarize_prof.py normalises profiled runs by this number)
    assert d["steps_executed"] >= d["spinup_steps_untimed"] + warmup + 2 * steps


def test_default_line(gpu_required):
    """`python bench.py` (inference headline, BASELINE.json configs[1]) with its secondary legs and the CPU baseline."""
    d = _run("--steps", "10", "--warmup", "2", "--min-leg-seconds", "0.05", "--sustained-seconds", "1.0")
    _check_common(d, 10, 2)
    su = d["sustained"]
    assert su["seconds"] >= 1.0 and su["steps"] > 10 and 0.7 * d["value"] < su["value"] < 1.2 * d["value"], su
    assert d["dtype"] == "f32" and "N=1024" in d["metric"] and d["config"]["num_points"] == 1024 and d["config"]["pairs_per_gpu"] == 256
    assert d["roofline"]["kernel"] == "pointnet_fused" and d["roofline"]["bound"] == "mfma" and d["roofline"]["frac"] > 0.5
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert d["train"]["dtype"] == "f32" and d["train"]["bf16"]["dtype"] == "bf16" and d["train"]["bf16"]["value"] > d["train"]["value"]
    # the short legs that put BASELINE.json configs[4] and the SURVEY 8(f) rows into the driver's record
    dgl = d["dgcnn"]
    assert dgl["infer"]["num_points"] == 4096 and dgl["infer"]["pairs_per_step"] == 512 and dgl["infer"]["value"] > 1000 and dgl["infer"]["kernel"].startswith("dgcnn_fused")
    assert dgl["train_f32"]["value"] > 100 and dgl["train_bf16"]["value"] > dgl["train_f32"]["value"] and dgl["train_bf16"]["last_train_kernel"] == 7
    assert d["loader"]["identical_batches"] is True and d["loader"]["packed_pairs_per_s"] > d["loader"]["file_based_pairs_per_s"] and d["loader"]["device_sampler_train"]["value"] > 1000
    assert d["icp"]["value"] > 100 and d["icp"]["mean_fitness"] > 0.5
    assert d["options"]["ab_mask"] == 0 and d["options"]["ablation_build"] is False and d["options"]["library"] == "libalignnet_hip.so"
    assert d["train"]["roofline"]["frac_lift_only"] < d["train"]["roofline"]["frac"]
    # `value` is timed under the per-kernel HIP-event timers (the contract's timed region); the line also carries the same K steps without them,
    # and the secondary training legs time their value without them and repeat the K steps under them for the roofline
    off = d["without_kernel_timers"]
    assert 0.95 * d["value"] <= off["value"] <= 1.10 * d["value"] and abs(off["value"] * off["ms_per_step"] * 1e-3 - 256) < 1.0, off
    for leg in (d["train"], d["train"]["bf16"]):
        assert 0.9 * leg["ms_per_step"] <= leg["ms_per_step_under_kernel_timers"] <= 1.25 * leg["ms_per_step"], leg
    assert d["pcie_inclusive"]["value"] < d["value"] * 1.05 and d["pcie_inclusive"]["pipelined"]["value"] > 0.9 * d["pcie_inclusive"]["value"] and d["infer_bf16x3"]["max_abs_diff_vs_exact_fp32_outputs"] < 1e-4


@pytest.mark.parametrize("args,kernel", [(("--mode", "train", "--train-dtype", "bf16"), "train_bwd_b2"),
                                         (("--workload", "dgcnn", "--batch", "32", "--points", "1024"), "dgcnn_fused"),
                                         (("--workload", "dgcnn", "--mode", "train", "--train-dtype", "bf16", "--batch", "256", "--points", "512"), "dg_train_bwd_edge")])   # (a full chip: below 256 clouds the edge kernels split clouds over workgroups and another pass may lead)
def test_other_lines(gpu_required, args, kernel):
    d = _run(*args, "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--sustained-seconds", "0")
    _check_common(d, 5, 1)
    assert d["roofline"]["kernel"].startswith(kernel), d["roofline"]["kernel"]
    if kernel == "train_bwd_b2":
        # pass B2's bf16 form is vector-instruction-issue-bound: quoted against the VALU issue rate, the matrix-pipe figure kept beside it
        r = d["roofline"]
        assert r["bound"] == "valu" and r["peak"] == 1228.8 and r["matrix_pipe"]["unit"] == "TFLOP/s" and r["valu_instr_per_launch"] > 0
        assert d["bn_mode"] == "local"
    assert d["steps_executed"] == d["spinup_steps_untimed"] + 1 + 2 * 5


@pytest.mark.parametrize("args", [("--mode", "train", "--train-dtype", "bf16", "--allreduce-overlap", "1"),
                                  ("--mode", "train", "--sync-bn", "1", "--allreduce-overlap", "0", "--grad-communicator", "1"),
                                  ("--min-leg-seconds", "0.05", "--no-split-leg", "--no-pcie-leg")])
def test_force_dist_rehearsal_at_world_1(gpu_required, args):
    """The code a --gpus N > 1 run takes -- bench.py re-launching itself under torch.distributed.run, init_process_group("nccl",
    device_id=...), the library's RCCL communicator (alignnet3d.parallel.init_comm), barriers, max over ranks, the per-rank gather, the
    sustained loop's flag all-reduce, the training headline and the training legs with both --allreduce-overlap values and
    --sync-bn 1 -- rehearsed at world = 1 on this box's one GPU (--force-dist), so that the driver's 8-GPU node is not its first
    execution.  What a 1-GPU box cannot show is xGMI itself: the scaling curve stays the driver's to measure."""
    d = _run("--force-dist", *args, "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--sustained-seconds", "0.6")
    _check_common(d, 5, 1)
    assert "forced_dist" in d and d["n_gpus"] == 1
    assert d["per_rank_pairs_per_s"]["ranks"] == 1 and abs(d["per_rank_pairs_per_s"]["min"] - d["value"]) <= 2e-2 * d["value"]
    assert d["sustained"]["steps"] > 5
    if "--mode" in args:
        assert "RCCL all-reduce of gradients, 1 ranks" in d["config"]["parallelism"] and "allreduce_exposed_ms_per_step" in d
        assert d["bn_mode"] == ("sync" if "--sync-bn" in args else "local")
        assert d["options"]["allreduce_overlap"] == int(args[args.index("--allreduce-overlap") + 1])
        assert d["options"]["grad_communicator"] == (1 if "--grad-communicator" in args else 0)
    else:
        for leg in (d["train"], d["train"]["bf16"]):
            assert leg["rccl_ranks"] == 1 and leg["bn_mode"] == "local" and leg["per_rank_pairs_per_s"]["ranks"] == 1, leg


def test_a_hung_secondary_leg_costs_the_leg_not_the_line(gpu_required):
    """Real ranks (torch.distributed.run): everything after the headline measurement -- the single-rank reference, the sustained loop, the training legs over
    RCCL -- runs under a watchdog (--secondary-timeout).  With a stand-in for a collective that never returns (BENCH_TEST_HANG_AFTER_HEADLINE) rank 0 still
    prints the one line, complete in its contract fields and marked `incomplete`, and the processes exit 0."""
    d = _run("--force-dist", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--secondary-timeout", "3", BENCH_TEST_HANG_AFTER_HEADLINE="1")
    _check_common(d, 5, 1)
    assert "watchdog" in d["incomplete"] and "train" not in d and "sustained" not in d
    assert d["roofline"]["frac"] > 0.5 and d["per_rank_pairs_per_s"]["ranks"] == 1


@pytest.mark.parametrize("args", [("--mode", "train", "--train-dtype", "bf16", "--sync-bn", "1"),
                                  ("--mode", "train", "--allreduce-overlap", "0"),
                                  ("--min-leg-seconds", "0.05", "--no-split-leg")])
def test_world_2_rehearsal_on_one_gpu(gpu_required, args):
    """bench.py's world > 1 code with TWO ranks on this box's one GPU (--rehearse-world 2: the ranks are host threads, the library's in-process loopback
    communicator stands where RCCL stands, a thread rendezvous where torch.distributed stands).  What the world-1 rehearsal cannot catch -- a
    collective that only rank 0 enters (round 5: the sync-BN latency probe of the training headline sat inside `if rank == 0`) -- hangs
    here until the rendezvous times out and fails the run.  Checks the line's multi-rank fields: per-rank gather over 2 ranks, the communicator's
    rank count, the exposed all-reduce, the single-rank reference the line scales from."""
    env = dict(os.environ, BENCH_REHEARSAL_TIMEOUT_S="120")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-world", "2", *args, "--steps", "5", "--warmup", "1", "--sustained-seconds", "0.3"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert "rehearsal" in d and d["n_gpus"] == 1 and d["config"]["devices_used"] == 1
    assert d["per_rank_pairs_per_s"]["ranks"] == 2
    ref = d["single_rank_reference"]
    assert ref["n_gpus"] == 1 and ref["value"] > 0 and abs(d["scaling_efficiency_vs_single_rank"] - d["value"] / (2 * ref["value"])) < 1e-3
    if "--mode" in args:
        assert "2 ranks" in d["config"]["parallelism"] and "allreduce_exposed_ms_per_step" in d
        if "--sync-bn" in args:
            assert d["bn_mode"] == "sync" and d["sync_collectives_per_step"] > 0 and d["sync_bn_latency_floor_ms"] > 0
    else:
        for leg in (d["train"], d["train"]["bf16"]):
            assert leg["rccl_ranks"] == 2 and leg["per_rank_pairs_per_s"]["ranks"] == 2, leg
