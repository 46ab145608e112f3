# PCIe-inclusive inference rate: host buffers in, host buffers out (alignnet_forward), B=256, N=1024.
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'alignnet-3d_amd')); sys.path.insert(0, ROOT)
import alignnet3d
from oracle import alignnet_ref as R
B, N = 256, 1024
eng = alignnet3d.Engine()
for name, shp, _ in eng.variables():
    if name.endswith("moving_var"): eng.set_variable(name, np.ones(shp[0] * shp[1], np.float32))
d = R.synth_pairs(B, N, dtype=np.float32)
for _ in range(3): eng.forward(d['pcs1'], d['pcs2'])
t = time.perf_counter(); K = 30
for _ in range(K): eng.forward(d['pcs1'], d['pcs2'])
dt = (time.perf_counter() - t) / K
print("PCIe-inclusive (pageable host buffers, blocking): %.3f ms/step, %.0f pairs/s" % (dt * 1e3, B / dt))
