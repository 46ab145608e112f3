# GPU: exact-fp32 vs split-bf16 backbone, B=256, N=1024 (device-resident inputs), same engine
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'alignnet-3d_amd')); sys.path.insert(0, ROOT)
import torch, alignnet3d
from oracle import alignnet_ref as R
B, N = 256, 1024
eng = alignnet3d.Engine()
d = R.synth_pairs(B, N, dtype=np.float32)
p1 = torch.tensor(d['pcs1']).cuda(); p2 = torch.tensor(d['pcs2']).cuda()
for mode in (0, 1, 0, 1):
    eng.set_option("infer_matmul_bf16x3", mode)
    for _ in range(5): eng.forward_device(p1.data_ptr(), p2.data_ptr(), B)
    eng.synchronize(); t = time.perf_counter(); K = 50
    for _ in range(K): eng.forward_device(p1.data_ptr(), p2.data_ptr(), B)
    eng.synchronize(); dt = (time.perf_counter() - t) / K
    print("split=%d  %.3f ms/step  %.0f pairs/s" % (mode, dt * 1e3, B / dt))
