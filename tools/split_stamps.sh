#!/bin/bash
# GPU, ablation build: cycle stamps of pointnet_split_persist's eight waves (workgroup 0, its third tile, C3 = 1024) at the phase boundaries
cd "$(dirname "$0")/.."
ALIGNNET_DBG=64 ALIGNNET_HIP_LIB=$PWD/alignnet-3d_amd/libalignnet_hip_ablate.so python - <<PY 2>&1 | grep "psp wave" | tail -16
import os, sys, numpy as np
sys.path[:0] = ["alignnet-3d_amd", "."]
import torch, alignnet3d
from alignnet3d.synth import synth_pairs
B, N = 256, 1024
eng = alignnet3d.Engine(); eng.set_option("infer_matmul_bf16x3", 1)
d = synth_pairs(B, N, seed=1, dtype=np.float32)
p1 = torch.tensor(d['pcs1']).cuda(); p2 = torch.tensor(d['pcs2']).cuda()
for _ in range(4): eng.forward_device(p1.data_ptr(), p2.data_ptr(), B)
eng.synchronize()
PY
