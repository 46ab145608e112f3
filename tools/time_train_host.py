# host enqueue time vs total time for the training step
import sys, time, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'alignnet-3d_amd')); sys.path.insert(0, ROOT)
import torch, alignnet3d
from oracle import alignnet_ref as R
B, N = 256, 1024
eng = alignnet3d.Engine()
d = R.synth_pairs(B, N, dtype=np.float32)
p1 = torch.tensor(d['pcs1']).cuda(); p2 = torch.tensor(d['pcs2']).cuda()
lab = {k: torch.tensor(np.ascontiguousarray(d[k])).cuda() for k in ("translations","rel_angles","pc1_centers","pc2_centers","pc1_angles","pc2_angles")}
lp = {k: v.data_ptr() for k, v in lab.items()}
for _ in range(3): eng.train_step_device(p1.data_ptr(), p2.data_ptr(), lp, B)
eng.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K): eng.train_step_device(p1.data_ptr(), p2.data_ptr(), lp, B)
t1 = time.perf_counter(); eng.synchronize(); t2 = time.perf_counter()
print("enqueue %.3f ms/step, total %.3f ms/step" % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
# single step latency (sync each)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); eng.train_step_device(p1.data_ptr(), p2.data_ptr(), lp, B); eng.synchronize(); ts.append(time.perf_counter() - t0)
print("single-step latency ms", [round(x * 1e3, 2) for x in ts])
