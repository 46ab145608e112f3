# conditioning check: the SINGLE engine on inputs moved by ~1 ulp -- how far does its own gradient move, per stage?
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "alignnet-3d_amd")]
import alignnet3d
from tests import test_loopback_gpu as L
N, B = 128, 16
for std in (True, False):
    cfg, spec, P32, d, du = L.setup("pointnet", N, B, std=std, seed=5)
    base = L.single_engine(cfg, P32, d, du, ())
    for eps in (1e-6, 1e-5):
        rng = np.random.default_rng(1)
        d2 = dict(d); d2["pcs1"] = (d["pcs1"] + eps * rng.standard_normal(d["pcs1"].shape)).astype(np.float32); d2["pcs2"] = (d["pcs2"] + eps * rng.standard_normal(d["pcs2"].shape)).astype(np.float32)
        pert = L.single_engine(cfg, P32, d2, du, ())
        stage = {}
        for n in base[1]:
            key = "s1" if "transformer1" in n else "s2" if "transformer2" in n else "s3"
            a, b = stage.setdefault(key, [0.0, 0.0])
            stage[key] = [a + float(np.sum((pert[1][n] - base[1][n]) ** 2)), b + float(np.sum(base[1][n] ** 2))]
        pd = max(float(np.abs(pert[0][k] - base[0][k]).max()) for k in ("pred_translations", "pred_remaining_angle_logits", "pred_s2_pc1centers"))
        print("std" if std else "non-std", "input noise %.0e: prediction change %.2e, gradient change per stage" % (eps, pd), {k: "%.2e" % np.sqrt(v[0] / v[1]) for k, v in sorted(stage.items())})
