#!/usr/bin/env python3
"""GPU: the engine against ITSELF with the inputs moved by one fp32 ulp, seed 13 (is the 1e-2 the engine's own conditioning?)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "alignnet-3d_amd")]
import alignnet3d
from oracle import alignnet_ref as R
from tests.helpers import small_cfg, oracle_params
from tests.test_fullsize_gpu import _grad_compare
W = dict(s1=(32, 64, 96), s2=(32, 64, 128), emb=(32, 64, 160))
for seed in (5, 13):
    N, B = 256, 16
    cfg = small_cfg(N=N, nb=12, fc=(64, 32), **W); cfg["training"]["batch_size"] = B
    spec, P32 = oracle_params(cfg, seed=seed)
    d = R.synth_pairs(B, N, seed=seed, dtype=np.float32)
    rng = np.random.default_rng(seed)
    du = {k: rng.uniform(size=(B, 32)).astype(np.float32) for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")}
    ul = [du[k] for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")]
    gs = []
    for pert in (0, 1, 2):
        dd = dict(d)
        if pert:
            r2 = np.random.default_rng(100 + pert)
            for k in ("pcs1", "pcs2"):
                dd[k] = np.nextafter(d[k], d[k] + np.where(r2.random(d[k].shape) < 0.5, -1, 1).astype(np.float32)).astype(np.float32)
        eng = alignnet3d.Engine(cfg); eng.set_variables(P32)
        res = eng.train_forward_backward(dd["pcs1"], dd["pcs2"], dd, ul)
        dec = eng.debug_train_decisions(B)
        gs.append(({n: eng.get_gradient(n).astype(np.float64) for n in R.trainable_names(spec)}, dec, res))
        eng.close()
    for i in (1, 2):
        _, relf, cos, rl2, _ = _grad_compare(gs[i][0].__getitem__, spec, gs[0][0])
        same = all((a == b).all() for a, b in zip(gs[i][1]["pool"], gs[0][1]["pool"])) and (gs[i][1]["yaw"] == gs[0][1]["yaw"]).all()
        ndiff = sum(int((a != b).sum()) for a, b in zip(gs[i][1]["pool"], gs[0][1]["pool"]))
        top = sorted(relf.items(), key=lambda kv: -kv[1])[:3]
        print("seed %d: engine vs engine(+-1 ulp inputs #%d): rl2 %.2e, decisions equal %s (%d pool winners differ), pred diff %.2e | %s" % (seed, i, rl2, same, ndiff,
              max(float(np.abs(gs[i][2][k] - gs[0][2][k]).max()) for k in alignnet3d.OUTPUT_NAMES), [(k, float("%.1e" % v)) for k, v in top]), flush=True)
