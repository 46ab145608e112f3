import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'alignnet-3d_amd')); sys.path.insert(0, ROOT)
import torch, alignnet3d
from oracle import alignnet_ref as R
B, N = 256, 1024
eng = alignnet3d.Engine(); eng.set_option("infer_matmul_bf16x3", 1)
d = R.synth_pairs(B, N, dtype=np.float32)
p1 = torch.tensor(d['pcs1']).cuda(); p2 = torch.tensor(d['pcs2']).cuda()
for _ in range(12): eng.forward_device(p1.data_ptr(), p2.data_ptr(), B)
eng.synchronize()
