#!/bin/bash
# GPU: where pointnet_split_persist's time goes -- the ablation build (make ablate) with parts of the kernel switched off (results are wrong, timing only):
# ALIGNNET_DBG 1 = last layer without its LDS reads, 2 = without its weight requests, 4 = no lift / hidden layer, 8 = no last layer.
cd "$(dirname "$0")/.."
for dbg in 0 1 2 3 4 8 12 7; do
  ALIGNNET_DBG=$dbg ALIGNNET_HIP_LIB=$PWD/alignnet-3d_amd/libalignnet_hip_ablate.so python - <<PY
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
for _ in range(30): eng.forward_device(p1.data_ptr(), p2.data_ptr(), B)
eng.synchronize(); ms, n = eng.profile_kernels()["backbone"]
print("dbg %2d: backbone %.3f ms/step (%s)" % ($dbg, ms / 30, eng.last_backbone_kernel()), flush=True)
PY
done
