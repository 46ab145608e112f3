#!/usr/bin/env python3
"""GPU: batch statistics of every BatchNorm layer, engine vs pinned fp64 oracle, seeds 5 / 13; smallest variances per layer."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "alignnet-3d_amd")]
import alignnet3d
from oracle import alignnet_ref as R
from tests import test_train_gpu as TT
from tests.helpers import small_cfg, oracle_params
W = dict(s1=(32, 64, 96), s2=(32, 64, 128), emb=(32, 64, 160))
for seed in (5, 13):
    N, B = 256, 16
    cfg = small_cfg(N=N, nb=12, fc=(64, 32), **W); cfg["training"]["batch_size"] = B
    spec, P32 = oracle_params(cfg, seed=seed)
    d = R.synth_pairs(B, N, seed=seed, dtype=np.float32)
    rng = np.random.default_rng(seed)
    du = {k: rng.uniform(size=(B, 32)).astype(np.float32) for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")}
    ul = [du[k] for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")]
    eng = alignnet3d.Engine(cfg); eng.set_variables(P32)
    decay = eng.state()["bn_decay"]
    res = eng.train_forward_backward(d["pcs1"], d["pcs2"], d, ul)
    dec = eng.debug_train_decisions(B)
    ep, loss, g, ema = TT._oracle(cfg, P32, d, du, decay, pinned=dec)
    print("seed", seed, "bn_decay", decay)
    for k in sorted(ema):
        if not k.endswith("moving_var"): continue
        got = eng.get_variable(k).astype(np.float64); ref = ema[k]
        batch_var = (ref - decay * P32[k]) / (1 - decay)
        km = k.replace("moving_var", "moving_mean")
        gm = eng.get_variable(km).astype(np.float64)
        print("  %-58s var: rel err %.1e, smallest batch var %.2e (largest %.2e) | mean abs err %.1e" % (k.replace("/moving_var", ""), np.abs(got - ref).max() / np.abs(ref).max(), batch_var.min(), batch_var.max(), np.abs(gm - ema[km]).max()))
    eng.close()
