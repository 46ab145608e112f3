#!/usr/bin/env python3
"""GPU: diagnosis of a gradient difference between the engine and the decision-pinned fp64 oracle (round 5: a 1e-2 difference at one seed of a
16 x 256 case turned out to be a relu sign within one rounding of zero -- DESIGN.md 2 viii).  Modes:
  tensors [N B seeds...]   per-tensor error, per seed and kernel-variant option (which tensor, which stage, which variant moves it)
  localise                 tower swap, batch subsets, other N / widths at seed 13 (does the error follow the data or a code path?)
  self                     the engine against ITSELF with its inputs moved by one ulp (same decisions?)
  statistics               every BatchNorm layer's batch statistics, engine vs oracle; smallest variances
Usage: python tools/pinned_diag.py MODE [args]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "alignnet-3d_amd")]
import alignnet3d
from oracle import alignnet_ref as R
from tests import test_train_gpu as TT
from tests.helpers import small_cfg, oracle_params
from tests.test_fullsize_gpu import _grad_compare
W = dict(s1=(32, 64, 96), s2=(32, 64, 128), emb=(32, 64, 160))


def tensors(ARGS):
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


def localise(ARGS):
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


def self_distance(ARGS):
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


def statistics(ARGS):
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


if __name__ == "__main__":
    {"tensors": tensors, "localise": localise, "self": self_distance, "statistics": statistics}[sys.argv[1]](sys.argv[2:])
