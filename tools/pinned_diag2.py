#!/usr/bin/env python3
"""GPU: where does the seed-13 gradient error come from?  Tower swap, batch subsets, N."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "alignnet-3d_amd")]
import alignnet3d
from oracle import alignnet_ref as R
from tests import test_train_gpu as TT
from tests.helpers import small_cfg, oracle_params
from tests.test_fullsize_gpu import _grad_compare
W = dict(s1=(32, 64, 96), s2=(32, 64, 128), emb=(32, 64, 160))
seed = 13
def run(tag, N, B, sel=None, swap=False, widths=W):
    cfg = small_cfg(N=N, nb=12, fc=(64, 32), **widths); cfg["training"]["batch_size"] = B if sel is None else len(sel)
    spec, P32 = oracle_params(cfg, seed=seed)
    d = R.synth_pairs(B, N, seed=seed, dtype=np.float32)
    rng = np.random.default_rng(seed)
    du = {k: rng.uniform(size=(B, 32)).astype(np.float32) for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")}
    if sel is not None:
        d = {k: v[sel] for k, v in d.items()}; du = {k: v[sel] for k, v in du.items()}
    if swap:
        P = {}
        for k, v in P32.items():
            k2 = k.replace("siamese_1/", "@@/").replace("siamese/", "siamese_1/").replace("@@/", "siamese/") if "/bn/" in k else k
            P[k2] = v
        P32 = P
        d = dict(d, pcs1=d["pcs2"], pcs2=d["pcs1"], pc1_centers=d["pc2_centers"], pc2_centers=d["pc1_centers"], pc1_angles=d["pc2_angles"], pc2_angles=d["pc1_angles"],
                 translations=-d["translations"], rel_angles=-d["rel_angles"])
        du = dict(du, s1_0=du["s1_1"], s1_1=du["s1_0"], s2_0=du["s2_1"], s2_1=du["s2_0"])
    ul = [du[k] for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")]
    eng = alignnet3d.Engine(cfg); eng.set_variables(P32)
    res = eng.train_forward_backward(d["pcs1"], d["pcs2"], d, ul)
    dec = eng.debug_train_decisions(d["pcs1"].shape[0])
    ge = {n: eng.get_gradient(n).astype(np.float64) for n in R.trainable_names(spec)}
    ep, loss, g, _ = TT._oracle(cfg, P32, d, du, eng.state()["bn_decay"], pinned=dec)
    eng.close()
    _, relf, cos, rl2, gs = _grad_compare(ge.__getitem__, spec, g)
    bn = {k: v for k, v in relf.items() if "/bn/" in k}
    t0 = max(v for k, v in bn.items() if k.startswith("siamese/")); t1 = max(v for k, v in bn.items() if k.startswith("siamese_1/"))
    top = sorted(relf.items(), key=lambda kv: -kv[1])[:3]
    print("%-34s rl2 %.2e | worst BN tensor tower0 %.1e tower1 %.1e | %s" % (tag, rl2, t0, t1, [(k.replace("siamese", "s").replace("transformer", "T").replace("embedding", "emb"), float("%.1e" % v)) for k, v in top]), flush=True)
run("base N=256 B=16", 256, 16)
run("towers swapped", 256, 16, swap=True)
run("first 8", 256, 16, sel=np.arange(8))
run("last 8", 256, 16, sel=np.arange(8, 16))
for lo in range(0, 16, 4):
    run("samples %d..%d" % (lo, lo + 3), 256, 16, sel=np.arange(lo, lo + 4))
run("N=128", 128, 16)
run("N=192", 192, 16)
run("std widths", 256, 16, widths=dict(s1=(64, 128, 96), s2=(64, 128, 128), emb=(64, 128, 160)))
