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
