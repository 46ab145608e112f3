#!/usr/bin/env python3
"""GPU: per-tensor gradient error of the engine against the decision-pinned fp64 oracle for one small case, over seeds and kernel-variant
options (which tensor, which stage, which variant moves it).  Usage: python tools/pinned_diag.py [N] [B] [seeds...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "alignnet-3d_amd")]
import alignnet3d
from oracle import alignnet_ref as R
from tests import test_train_gpu as TT
from tests.helpers import small_cfg, oracle_params
from tests.test_fullsize_gpu import _grad_compare
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
seeds = [int(x) for x in sys.argv[3:]] or [5, 13, 21]
W = dict(s1=(32, 64, 96), s2=(32, 64, 128), emb=(32, 64, 160))
for seed in seeds:
    cfg = small_cfg(N=N, nb=12, fc=(64, 32), **W); cfg["training"]["batch_size"] = B
    spec, P32 = oracle_params(cfg, seed=seed)
    d = R.synth_pairs(B, N, seed=seed, dtype=np.float32)
    rng = np.random.default_rng(seed)
    du = {k: rng.uniform(size=(B, 32)).astype(np.float32) for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")}
    ul = [du[k] for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")]
    base = None
    for opts in ((), (("ab_no_glue_fold", 1),), (("train_phase3_tile64", 1),)):
        eng = alignnet3d.Engine(cfg); eng.set_variables(P32)
        for k, v in opts: eng.set_option(k, v)
        res = eng.train_forward_backward(d["pcs1"], d["pcs2"], d, ul)
        dec = eng.debug_train_decisions(B)
        ge = {n: eng.get_gradient(n).astype(np.float64) for n in R.trainable_names(spec)}
        if base is None:
            ep, loss, g, _ = TT._oracle(cfg, P32, d, du, eng.state()["bn_decay"], pinned=dec)
            base = (ep, loss, g)
        eng.close()
        _, relf, cos, rl2, gs = _grad_compare(ge.__getitem__, spec, base[2])
        top = sorted(relf.items(), key=lambda kv: -kv[1])[:6]
        if not opts:
            for k, v in sorted(relf.items()):
                print("      %-60s %.2e  |ref|max %.2e" % (k, v, np.abs(base[2][k]).max()))
        print("seed %d %-60s rl2 %.2e | %s" % (seed, dict(opts) or "default", rl2, [(k.replace("siamese", "s").replace("transformer", "T").replace("embedding", "emb"), float("%.1e" % v)) for k, v in top]), flush=True)
