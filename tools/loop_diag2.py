# diagnostic: which kernel variant makes the sharded step differ from the single engine (std widths, fp32)
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "alignnet-3d_amd")]
import alignnet3d
from tests import test_loopback_gpu as L
N, B, W = 128, 16, 2
cfg, spec, P32, d, du = L.setup("pointnet", N, B, std=True, seed=5)
for opts in [(), (("ab_p3_nogram", 1),), (("train_phase3_tile64", 1),), (("ab_no_glue_fold", 1),), (("ab_no_ld_const", 1),)]:
    single = L.single_engine(cfg, P32, d, du, opts)
    ranks = L.sharded_step(W, cfg, P32, d, du, opts)
    gs, gr = single[1], ranks[0]["summed"]
    stage = {}
    for n in gs:
        key = "s1" if "transformer1" in n else "s2" if "transformer2" in n else "s3"
        a, b = stage.setdefault(key, [0.0, 0.0])
        stage[key] = [a + float(np.sum((gr[n] - gs[n]) ** 2)), b + float(np.sum(gs[n] ** 2))]
    print(opts, {k: "%.2e" % np.sqrt(v[0] / v[1]) for k, v in sorted(stage.items())})
