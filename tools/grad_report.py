import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/alignnet-3d_amd')
import alignnet3d
from oracle import alignnet_ref as R
from tests.test_train_gpu import _setup, _oracle
N, B = int(sys.argv[1]), int(sys.argv[2])
cfg, spec, P32, d, du = _setup(N, B)
eng = alignnet3d.Engine(cfg); eng.set_variables(P32)
_, _, grads, _ = _oracle(cfg, P32, d, du, eng.state()["bn_decay"])
eng.train_forward_backward(d["pcs1"], d["pcs2"], d, [du[k] for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")])
rows = []
for name in R.trainable_names(spec):
    g = eng.get_gradient(name).astype(np.float64).ravel(); ref = grads[name].ravel()
    rows.append((np.abs(g - ref).max() / (np.abs(ref).max() + 1e-30), np.abs(g - ref).max(), np.abs(ref).max(), name))
rows = [r for r in rows if r[2] > 1e-9]; rows.sort(reverse=True)
for r in rows[:14]: print("rel %.2e abs %.2e refmax %.2e %s" % r)
