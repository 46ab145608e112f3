# whole-train-step wall time (ablation helper; ALIGNNET_DBG flags skip parts of the kernels)
import sys, time, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'alignnet-3d_amd')); sys.path.insert(0, ROOT)
import torch, alignnet3d
from oracle import alignnet_ref as R
B, N = int(os.environ.get('ALIGNNET_B', 256)), int(os.environ.get('ALIGNNET_N', 1024))
cfg = alignnet3d.default_model_config()
cfg['model']['num_points'] = N
if os.environ.get('ALIGNNET_DGCNN'): cfg['model']['backbone'] = 'dgcnn'
eng = alignnet3d.Engine(cfg)
if os.environ.get('ALIGNNET_BF16'): eng.set_option('train_matmul_bf16', 1)
d = R.synth_pairs(B, N, dtype=np.float32)
p1 = torch.tensor(d['pcs1']).cuda(); p2 = torch.tensor(d['pcs2']).cuda()
lab = {k: torch.tensor(np.ascontiguousarray(d[k])).cuda() for k in ("translations","rel_angles","pc1_centers","pc2_centers","pc1_angles","pc2_angles")}
lp = {k: v.data_ptr() for k, v in lab.items()}
K = int(os.environ.get('ALIGNNET_K', 8))
for _ in range(2 if K <= 8 else K): eng.train_step_device(p1.data_ptr(), p2.data_ptr(), lp, B)
eng.synchronize(); t = time.time()
for _ in range(K): eng.train_step_device(p1.data_ptr(), p2.data_ptr(), lp, B)
eng.synchronize(); print("DBG=%s BF16=%s ms/step %.3f" % (os.environ.get("ALIGNNET_DBG", "0"), os.environ.get("ALIGNNET_BF16", "0"), (time.time() - t) / K * 1e3))
