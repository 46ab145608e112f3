"""GPU: HBM-resident dataset + device-side batch sampler (include/alignnet_hip.h alignnet_dataset_*), the replacement of
provider.load_batch + jitter_point_cloud (reference provider.py:60-71,85-136, train.py:352-356).
Oracle: oracle/dataset_ref.py (same counter hash in NumPy).  Bars: gather indices / un-jittered points / labels
bit-exact; jittered points within 2e-6 of the oracle (device vs host logf/cosf), never beyond the clip."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import alignnet3d
from oracle import alignnet_ref as R
from oracle import dataset_ref as D
from tests.helpers import small_cfg, oracle_params
from tests.test_dropin_gpu import _make_dataset, _run, PKG, ROOT

pytestmark = pytest.mark.gpu
LABELS = ("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles")
WIDTH = dict(translations=3, rel_angles=1, pc1_centers=3, pc2_centers=3, pc1_angles=1, pc2_angles=1)


def _toy(n=9, seed=3):
    rng = np.random.default_rng(seed)
    cnt = rng.integers(1, 90, (n, 2))
    cnt[4, 1] = 0                       # an empty cloud samples as zeros (provider.py:97-98)
    cnt[5, 0] = 7                       # small cloud for the uniformity check
    off = np.zeros((n + 1, 2), np.int64)
    off[1:] = np.cumsum(cnt, 0)
    pts = [rng.normal(size=(off[-1, t], 3)).astype(np.float32) * 3 for t in range(2)]
    lab = rng.normal(size=(n, 12)).astype(np.float32)
    return pts, off, lab


def _read_batch(eng, ptrs, B, N):
    p1, p2, L = ptrs
    a = eng.read_device(p1, B * N * 3).reshape(B, N, 3)
    b = eng.read_device(p2, B * N * 3).reshape(B, N, 3)
    labs = {k: eng.read_device(L[k], B * WIDTH[k]).reshape(B, WIDTH[k]) for k in LABELS}
    return a, b, labs


@pytest.mark.parametrize("N", [64, 300])
def test_sampler_matches_oracle(gpu_required, N):
    cfg = small_cfg(N=N, nb=12)
    eng = alignnet3d.Engine(cfg)
    pts, off, lab = _toy()
    with pytest.raises(RuntimeError):
        eng.sample_batch([0, 1], seed=1)          # nothing uploaded yet
    eng.upload_dataset(pts[0], pts[1], off, lab)
    rows = [0, 4, 5, 8, 4, 2]
    # no jitter: pure gather, bit-exact
    a, b, labs = _read_batch(eng, eng.sample_batch(rows, seed=77), len(rows), N)
    ra, rb, rl, picks = D.sample_batch(pts, off, lab, rows, N, 77)
    assert np.array_equal(a, ra) and np.array_equal(b, rb)
    for k in LABELS:
        assert np.array_equal(labs[k], rl[k]), k
    assert not b[1].any() and not b[4].any()       # the empty cloud
    # every sampled point is a point of its source cloud; rows repeated in a batch draw the same sample
    src = pts[0][off[5, 0]:off[6, 0]]
    assert all((src == p).all(1).any() for p in a[2])
    assert np.array_equal(a[1], a[4])
    # a different seed draws a different sample; the same seed reproduces it
    a2, _, _ = _read_batch(eng, eng.sample_batch(rows, seed=78), len(rows), N)
    a3, _, _ = _read_batch(eng, eng.sample_batch(rows, seed=77), len(rows), N)
    assert not np.array_equal(a2, a) and np.array_equal(a3, a)
    # jitter: clip(sigma * N(0,1), +-clip) added to the gathered point
    sigma, clip = 0.01, 0.05
    j, jb, _ = _read_batch(eng, eng.sample_batch(rows, seed=77, jitter_sigma=sigma, jitter_clip=clip), len(rows), N)
    rj, rjb, _, _ = D.sample_batch(pts, off, lab, rows, N, 77, sigma, clip)
    np.testing.assert_allclose(j, rj, rtol=0, atol=2e-6)
    np.testing.assert_allclose(jb, rjb, rtol=0, atol=2e-6)
    assert np.abs(j - a).max() <= clip + 1e-6
    with pytest.raises(RuntimeError):
        eng.sample_batch([0, 99], seed=1)           # row out of range
    with pytest.raises(RuntimeError):
        eng.sample_batch([0], seed=1, jitter_sigma=0.01, jitter_clip=0.0)   # provider.py:68 assert (clip > 0)
    eng.close()


def test_sampler_distribution(gpu_required):
    """np.random.choice(n, N, replace=True) is uniform over the cloud; randn jitter has std sigma (before clipping at 5 sigma)."""
    N = 4096
    eng = alignnet3d.Engine(small_cfg(N=N, nb=12))
    pts, off, lab = _toy()
    eng.upload_dataset(pts[0], pts[1], off, lab)
    counts = np.zeros(7)
    for seed in range(8):
        a, _, _ = _read_batch(eng, eng.sample_batch([5], seed=seed), 1, N)
        src = pts[0][off[5, 0]:off[6, 0]]
        counts += (a[0][:, None, :] == src[None]).all(2).sum(0)
    expect = 8 * N / 7
    chi2 = float(((counts - expect) ** 2 / expect).sum())
    assert chi2 < 22.5, (chi2, counts)              # chi-square, 6 dof, p = 0.001
    clean, _, _ = _read_batch(eng, eng.sample_batch([0, 1, 2], seed=5), 3, N)
    jit, _, _ = _read_batch(eng, eng.sample_batch([0, 1, 2], seed=5, jitter_sigma=0.01, jitter_clip=0.05), 3, N)
    d = (jit - clean).ravel()
    assert abs(d.std() - 0.01) < 3e-4 and abs(d.mean()) < 3e-4, (d.std(), d.mean())
    z = np.sort(d / 0.01)
    q = np.array([0.1, 0.25, 0.5, 0.75, 0.9])
    np.testing.assert_allclose(z[(q * z.size).astype(int)], [-1.2816, -0.6745, 0.0, 0.6745, 1.2816], atol=0.03)
    eng.close()


def test_rows_entry_points_match_host_path(gpu_required):
    """forward_rows / train_step_rows on the sampled batch == forward / train_step fed the same batch from the host."""
    N, B = 128, 6
    cfg = small_cfg(N=N, nb=12, s1=(32, 64, 96), s2=(32, 64, 128), emb=(32, 64, 160), fc=(64, 32))
    cfg["training"]["batch_size"] = B
    spec, P32 = oracle_params(cfg, seed=5)
    pts, off, lab = _toy()
    rows = [0, 1, 2, 3, 6, 7]
    out = []
    for mode in ("rows", "host"):
        eng = alignnet3d.Engine(cfg)
        eng.set_variables(P32)
        eng.upload_dataset(pts[0], pts[1], off, lab)
        if mode == "rows":
            ep = eng.forward_rows(rows, seed=9)
            res = eng.train_step_rows(rows, seed=11, jitter_sigma=0.01, jitter_clip=0.05)
        else:
            a, b, _ = _read_batch(eng, eng.sample_batch(rows, seed=9), B, N)
            ep = eng.forward(a, b)
            a, b, labs = _read_batch(eng, eng.sample_batch(rows, seed=11, jitter_sigma=0.01, jitter_clip=0.05), B, N)
            res = eng.train_step(a, b, labs)
        out.append((ep, res["loss"], eng.get_variable("fc3/weights")))
        eng.close()
    for k in out[0][0]:
        assert np.array_equal(out[0][0][k], out[1][0][k]), k
    assert out[0][1] == out[1][1]
    assert np.array_equal(out[0][2], out[1][2])


def test_train_py_device_dataset(gpu_required, tmp_path):
    """train.py with ALIGNNET_DEVICE_DATASET=1 (+ the bf16 lift option): same files written, finite losses, no per-step host batches."""
    root = tmp_path / "SynthTiny"
    _make_dataset(str(root))
    user = {"data": {"basepath": str(root)}, "logging": {"basedir": str(tmp_path / "logs")},
            "model": {"num_points": 64, "angles": {"num_bins": 12, "accept_inverted_angle": True},
                      "options": {"s1transformer": [[32, 64, 96], [[64, 32], 0.7]], "s2transformer": [[32, 64, 128], [[64, 32], 0.7]],
                                  "embedding": [32, 64, 160], "remaining_transform_prediction": [[64, 32], 0.7]}},
            "training": {"batch_size": 8, "num_epochs": 2, "learning_rate": 0.002}}
    cfgp = tmp_path / "DevRun.json"
    json.dump(user, open(cfgp, "w"))
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + ROOT, ALIGNNET_DEVICE_DATASET="1", ALIGNNET_TRAIN_BF16="1")
    r = subprocess.run([sys.executable, os.path.join(PKG, "train.py"), "train", "--config", str(cfgp)], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = r.stdout + r.stderr
    logdir = tmp_path / "logs" / "DevRun"
    assert "train mean loss" in out and "Finished Training" in out and "nan" not in out.lower()
    assert "bf16 operands" in out
    ev = logdir / "val" / "eval000001"
    assert (ev / "pred_translations.npy").exists() and (ev / "eval.json").exists()
    assert np.isfinite(np.load(ev / "pred_translations.npy")).all()
    assert (root / "packed_cache" / "ids.npy").exists()
    # eval_only --refineICP (train.py:401-402,463-484): refined results next to the plain ones, rotation centre reset to 0
    env.pop("ALIGNNET_DEVICE_DATASET")
    for extra, sub in ((["--refineICP"], "refined_p2p"), (["--refineICP", "--its", "5", "--use_old_results"], "refined_p2p_5")):
        r = subprocess.run([sys.executable, os.path.join(PKG, "train.py"), "eval_only", "--config", str(cfgp), "--eval_epoch", "1"] + extra,
                           cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        ref = ev / sub
        t, a, c = (np.load(ref / ("%s.npy" % k)) for k in ("pred_translations", "pred_angles", "pred_s2_pc1centers"))
        assert np.isfinite(t).all() and np.isfinite(a).all() and not c.any() and (ref / "eval.json").exists()
        assert t.shape == np.load(ev / "pred_translations.npy").shape
