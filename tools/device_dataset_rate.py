# GPU: training fed by (a) host batches drawn with the packed loader's NumPy resampling + jitter + H2D copy per step,
# (b) the HBM-resident dataset + device sampler (alignnet_dataset_*).  Synthetic 2048-example dataset, B=256, N=1024.
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'alignnet-3d_amd')); sys.path.insert(0, ROOT)
import alignnet3d
from oracle import alignnet_ref as R
n, B, N, steps = 2048, 256, 1024, 12
d = R.synth_pairs(n, 1500, dtype=np.float32)
off = np.zeros((n + 1, 2), np.int64); off[1:, 0] = off[1:, 1] = np.arange(1, n + 1) * 1500
p1, p2 = d["pcs1"].reshape(-1, 3), d["pcs2"].reshape(-1, 3)
lab = np.concatenate([d["translations"], d["rel_angles"], d["pc1_centers"], d["pc2_centers"], d["pc1_angles"], d["pc2_angles"]], 1).astype(np.float32)
eng = alignnet3d.Engine()
t = time.perf_counter(); eng.upload_dataset(p1, p2, off, lab); eng.synchronize(); t_up = time.perf_counter() - t
rng = np.random.default_rng(0)
def host_batch(rows):
    a = np.empty((B, N, 3)); b = np.empty((B, N, 3))
    for r, i in enumerate(rows):   # provider.py:97-98 + :60-71 semantics (float64 like the reference loader)
        a[r] = p1[off[i, 0] + np.random.choice(1500, N, replace=True)]
        b[r] = p2[off[i, 1] + np.random.choice(1500, N, replace=True)]
    a += np.clip(0.01 * np.random.randn(B, N, 3), -0.05, 0.05); b += np.clip(0.01 * np.random.randn(B, N, 3), -0.05, 0.05)
    L = lab[rows]
    return a, b, dict(translations=L[:, 0:3], rel_angles=L[:, 3:4], pc1_centers=L[:, 4:7], pc2_centers=L[:, 7:10], pc1_angles=L[:, 10:11], pc2_angles=L[:, 11:12])
for mode in ("host", "device"):
    for k in range(steps + 2):
        if k == 2: eng.synchronize(); t0 = time.perf_counter()
        rows = rng.integers(0, n, B)
        if mode == "host":
            a, b, L = host_batch(rows); eng.train_step(a, b, L)
        else:
            eng.train_step_rows(rows, seed=k)
    eng.synchronize(); dt = (time.perf_counter() - t0) / steps
    print("%-6s fed training: %.2f ms/step = %.0f pairs/s" % (mode, dt * 1e3, B / dt))
print("one-off upload of %d examples (%.0f MB): %.2f s" % (n, (p1.nbytes + p2.nbytes) / 1e6, t_up))
