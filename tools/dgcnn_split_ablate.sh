#!/bin/bash
# GPU, ablation build: where dgcnn_split's time goes (N = 4096, 128 pairs) -- ALIGNNET_DBG 1 = no edge conv MFMAs, 2 = no lift, 4 = no point conv (results wrong, timing only)
cd "$(dirname "$0")/.."
for dbg in 0 1 2 3 4 7; do
  ALIGNNET_DBG=$dbg ALIGNNET_HIP_LIB=$PWD/alignnet-3d_amd/libalignnet_hip_ablate.so python - <<PY 2>&1 | grep dbg
import os, sys, time, numpy as np
sys.path[:0] = ["alignnet-3d_amd", "."]
import torch, alignnet3d
from alignnet3d.synth import synth_pairs
B, N = 128, 4096
cfg = alignnet3d.default_model_config(); cfg["model"]["backbone"] = "dgcnn"; cfg["model"]["num_points"] = N
eng = alignnet3d.Engine(cfg); eng.set_option("infer_matmul_bf16x3", 1)
d = synth_pairs(B, N, seed=1, dtype=np.float32)
p1 = torch.tensor(d['pcs1']).cuda(); p2 = torch.tensor(d['pcs2']).cuda()
for _ in range(2): eng.forward_device(p1.data_ptr(), p2.data_ptr(), B)
eng.synchronize(); eng.profile_enable(True); eng.profile_read(reset=True)
for _ in range(6): eng.forward_device(p1.data_ptr(), p2.data_ptr(), B)
eng.synchronize(); k = eng.profile_kernels()
print("dbg %2d: backbone %.3f ms/step, knn %.3f (%s)" % ($dbg, k["backbone"][0] / 6, k["knn"][0] / 6, eng.last_backbone_kernel()), flush=True)
PY
done
