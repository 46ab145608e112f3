# per-tensor gradient error of the general-depth training path vs the fp64 autograd oracle, next to the fp32 oracle's own error
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "alignnet-3d_amd")]
import numpy as np
import alignnet3d
from oracle import alignnet_ref as R
from tests.helpers import small_cfg, oracle_params
from tests import test_train_gpu as TT
case, N, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = small_cfg(N=N, nb=12, fc=(64, 32), **TT.GENERAL_DEPTH[case]); cfg["training"]["batch_size"] = B
spec, P32 = oracle_params(cfg, seed=9)
d = R.synth_pairs(B, N, seed=9, dtype=np.float32)
rng = np.random.default_rng(9)
du = {k: rng.uniform(size=(B, 32)).astype(np.float32) for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")}
eng = alignnet3d.Engine(cfg); eng.set_variables(P32)
e64, l64, g64, _ = TT._oracle(cfg, P32, d, du, 0.5)
e32, l32, g32, _ = TT._oracle(cfg, P32, d, du, 0.5, dt=np.float32)
eng.train_forward_backward(d["pcs1"], d["pcs2"], d, [du[k] for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")])
gs = max(float(np.abs(v).max()) for v in g64.values())
for n, _ in R.param_names(spec):
    if n not in g64 or n.endswith("biases"): continue
    g = eng.get_gradient(n).astype(np.float64).reshape(g64[n].shape)
    den = float(np.abs(g64[n]).max()) + 1e-5 * gs
    print("%-58s hip %.2e   fp32-oracle %.2e   |g|max %.2e" % (n, np.abs(g - g64[n]).max() / den, np.abs(g32[n] - g64[n]).max() / den, np.abs(g64[n]).max()))
t0 = max(np.abs(eng.get_gradient(n).astype(np.float64).reshape(g64[n].shape) - g64[n]).max() / (np.abs(g64[n]).max() + 1e-5 * gs) for n in g64 if n.startswith("siamese/") and "/bn/" in n)
t1 = max(np.abs(eng.get_gradient(n).astype(np.float64).reshape(g64[n].shape) - g64[n]).max() / (np.abs(g64[n]).max() + 1e-5 * gs) for n in g64 if n.startswith("siamese_1/") and "/bn/" in n)
print("SUMMARY", case, N, B, "worst BN-grad error tower0 %.2e tower1 %.2e" % (t0, t1))
