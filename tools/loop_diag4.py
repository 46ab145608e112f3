import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "alignnet-3d_amd")]
import alignnet3d
from oracle import alignnet_ref as R
from tests import test_loopback_gpu as L
from tests.helpers import oracle_params
for N in (1024, 128):
    cfg = alignnet3d.default_model_config(); cfg["model"]["num_points"] = N; cfg["training"]["batch_size"] = 2048
    spec, P32 = oracle_params(cfg, seed=11)
    d = R.synth_pairs(2048, N, seed=11, dtype=np.float32)
    rng = np.random.default_rng(11); du = {k: rng.uniform(size=(2048, 256)).astype(np.float32) for k in L.U}
    for bf16 in (0, 1):
        s = L.single_engine(cfg, P32, d, du, (("train_matmul_bf16", bf16),))
        m = L.yaw_margin(s[0], 50)
        print(N, "bf16" if bf16 else "fp32", "margin quantiles", np.quantile(m, [0.001, 0.01, 0.05, 0.25, 0.5]).round(5), "below 1e-3: %d, 5e-3: %d, 3e-2: %d" % ((m < 1e-3).sum(), (m < 5e-3).sum(), (m < 3e-2).sum()))
