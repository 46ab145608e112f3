#!/bin/bash
# GPU, ablation build: pointnet_split_persist's priority pattern (SplitArgs::prio_mask, ALIGNNET_DBG bits 8 .. 15) against the backbone's time
cd "$(dirname "$0")/.."
for mask in 0x55 0x5b 0x6d 0x77 0xdb 0x7f 0xff 0x11 0x01; do
  ALIGNNET_DBG=$(( (mask << 8) )) ALIGNNET_HIP_LIB=$PWD/alignnet-3d_amd/libalignnet_hip_ablate.so python - <<PY 2>&1 | grep mask
import os, sys, time, numpy as np
sys.path[:0] = ["alignnet-3d_amd", "."]
import torch, alignnet3d
from alignnet3d.synth import synth_pairs
B, N = 256, 1024
eng = alignnet3d.Engine(); eng.set_option("infer_matmul_bf16x3", 1)
d = synth_pairs(B, N, seed=1, dtype=np.float32)
p1 = torch.tensor(d['pcs1']).cuda(); p2 = torch.tensor(d['pcs2']).cuda()
for _ in range(5): eng.forward_device(p1.data_ptr(), p2.data_ptr(), B)
eng.synchronize(); eng.profile_enable(True); eng.profile_read(reset=True)
for _ in range(40): eng.forward_device(p1.data_ptr(), p2.data_ptr(), B)
eng.synchronize(); ms, n = eng.profile_kernels()["backbone"]
print("mask %s: backbone %.4f ms/step (%s)" % ("$mask", ms / 40, eng.last_backbone_kernel()), flush=True)
PY
done
