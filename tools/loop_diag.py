# diagnostic: sharded (loopback) vs single engine vs fp64 oracle, per tensor
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "alignnet-3d_amd")]
import alignnet3d
from oracle import alignnet_ref as R
from tests import test_loopback_gpu as L
from tests import test_train_gpu as TT
backbone = sys.argv[1] if len(sys.argv) > 1 else "pointnet"
std = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W = int(sys.argv[3]) if len(sys.argv) > 3 else 2
N, B = 128, 16
cfg, spec, P32, d, du = L.setup(backbone, N, B, std=bool(std), seed=7 if backbone == "dgcnn" else 5)
ep, loss, go, ema = TT._oracle(cfg, P32, d, du, 0.5)
single = L.single_engine(cfg, P32, d, du, ())
ranks = L.sharded_step(W, cfg, P32, d, du, ())
gs = single[1]; gr = ranks[0]["summed"]
print("loss oracle %.7f single %.7f sharded %.7f" % (loss, single[0]["loss"], ranks[0]["res"]["loss"]))
rows = []
for n in gs:
    o = np.asarray(go[n], np.float64).reshape(gs[n].shape)
    den = np.linalg.norm(o) + 1e-12
    rows.append((n, np.linalg.norm(gs[n] - o) / den, np.linalg.norm(gr[n] - o) / den, np.linalg.norm(gr[n] - gs[n]) / den, np.abs(o).max()))
rows.sort(key=lambda r: -r[3])
print("%-60s %10s %10s %10s %10s" % ("tensor", "single-orc", "shard-orc", "shard-sing", "max|g|"))
for r in rows[:25]:
    print("%-60s %10.2e %10.2e %10.2e %10.2e" % r)
