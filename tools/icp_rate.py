# GPU: ICP refinement throughput (alignnet_icp_refine_dataset), 256 pairs x 1500-point clouds, radius 0.1, 30 iterations,
# seeded near the truth like the network's prediction; the NumPy oracle beside it on a few pairs.
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'alignnet-3d_amd')); sys.path.insert(0, ROOT)
import alignnet3d
from oracle import alignnet_ref as R
from oracle import icp_ref as I
n, P = 256, 1500
d = R.synth_pairs(n, P, dtype=np.float32)
off = np.zeros((n + 1, 2), np.int64); off[1:, 0] = off[1:, 1] = np.arange(1, n + 1) * P
eng = alignnet3d.Engine()
eng.upload_dataset(d["pcs1"].reshape(-1, 3), d["pcs2"].reshape(-1, 3), off, np.zeros((n, 12), np.float32))
rng = np.random.default_rng(0)
inits = [I.get_mat_angle(d["translations"][i] + rng.normal(0, 0.05, 3), float(d["rel_angles"][i, 0]) + rng.normal(0, 0.03), rotation_center=d["pc1_centers"][i]) for i in range(n)]
rows = np.arange(n)
eng.icp_refine_rows(rows, inits, 0.1, 30)
t = time.perf_counter(); res = eng.icp_refine_rows(rows, inits, 0.1, 30); dt = time.perf_counter() - t
print("GPU: %d pairs in %.1f ms = %.0f pairs/s (mean iterations %.1f, mean fitness %.2f)" % (n, dt * 1e3, n / dt, res["iterations"].mean(), res["fitness"].mean()))
t = time.perf_counter()
for i in range(4): I.icp_p2point_z(d["pcs1"][i], d["pcs2"][i], inits[i], 0.1, 30)
dc = (time.perf_counter() - t) / 4
print("NumPy oracle: %.1f ms per pair = %.1f pairs/s" % (dc * 1e3, 1 / dc))
