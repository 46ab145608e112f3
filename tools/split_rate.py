#!/usr/bin/env python3
"""GPU: the split-bf16 PointNet backbone (option infer_matmul_bf16x3) at B = 256, N = 1024, device-resident inputs: the persistent kernel
(pointnet_split_persist, default) against the one-workgroup-per-tile kernel (ab_split_tilewise = 1) and the exact-fp32 path, alternating on one
engine; step time from the wall clock (kernel timers off) and the backbone's time per step from the kernel timers.
Usage: python tools/split_rate.py [rounds]"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'alignnet-3d_amd')); sys.path.insert(0, ROOT)
import torch, alignnet3d
from alignnet3d.synth import synth_pairs
B, N = 256, 1024
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
eng = alignnet3d.Engine()
d = synth_pairs(B, N, seed=1, dtype=np.float32)
p1 = torch.tensor(d['pcs1']).cuda(); p2 = torch.tensor(d['pcs2']).cuda()
MODES = (("exact fp32", 0, 0), ("split tilewise", 1, 1), ("split persistent", 1, 0))
for r in range(rounds):
    for name, split, tilewise in MODES:
        eng.set_option("infer_matmul_bf16x3", split); eng.set_option("ab_split_tilewise", tilewise)
        for _ in range(5): eng.forward_device(p1.data_ptr(), p2.data_ptr(), B)
        eng.synchronize(); t = time.perf_counter(); K = 100
        for _ in range(K): eng.forward_device(p1.data_ptr(), p2.data_ptr(), B)
        eng.synchronize(); dt = (time.perf_counter() - t) / K
        eng.profile_enable(True); eng.profile_read(reset=True)
        for _ in range(20): eng.forward_device(p1.data_ptr(), p2.data_ptr(), B)
        eng.synchronize(); kern = eng.profile_kernels(); eng.profile_read(reset=True); eng.profile_enable(False)
        ms, n = kern["backbone"]
        print("%-17s %.3f ms/step  %7.0f pairs/s | backbone %.3f ms/step in %d launches (%s)" % (name, dt * 1e3, B / dt, ms / 20, n // 20, eng.last_backbone_kernel()), flush=True)
