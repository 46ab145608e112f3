import os, sys, numpy as np
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "alignnet-3d_amd")]
import alignnet3d
from oracle import alignnet_ref as R
from tests.test_train_gpu import _setup_dgcnn
cfg, spec, P32, d, du = _setup_dgcnn(128, 8, std=False)
ul = [du[k] for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")]
eng = alignnet3d.Engine(cfg); eng.set_variables(P32)
res = eng.train_forward_backward(d["pcs1"], d["pcs2"], d, ul)
g = {n: eng.get_gradient(n).copy() for n in R.trainable_names(spec)}
np.savez(sys.argv[1], loss=res["loss"], **{n.replace("/", "__"): v for n, v in g.items()})
