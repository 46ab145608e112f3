import sys, time, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'alignnet-3d_amd')); sys.path.insert(0, ROOT)
import alignnet3d
from oracle import alignnet_ref as R
B, N = int(sys.argv[1]), int(sys.argv[2])
cfg = alignnet3d.default_model_config(); cfg["model"]["num_points"] = N
eng = alignnet3d.Engine(cfg)
d = R.synth_pairs(B, N, dtype=np.float32)
t = time.time(); r = eng.train_step(d['pcs1'], d['pcs2'], d); print("B", B, "N", N, "first step %.2fs loss %.4f" % (time.time() - t, r['loss']), flush=True)
t = time.time(); r = eng.train_step(d['pcs1'], d['pcs2'], d); print("second step %.4fs" % (time.time() - t), flush=True)
