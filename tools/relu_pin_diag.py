#!/usr/bin/env python3
"""GPU: where does the engine's distance from the fully pinned fp64 oracle come from?  One training step; the oracle pinned to the engine's
decisions AND relu signs (a smooth function of its inputs) is evaluated in fp64 and -- the same torch graph, the same pins -- in fp32: the
second is what a straightforward fp32 evaluation of the reference graph gives on this batch.  Prints whole-gradient relative L2 of
engine-vs-fp64 and torch-fp32-vs-fp64 and the worst tensors of each (error over max(own largest entry, 2 % of the gradient's)).
Usage: python tools/relu_pin_diag.py CASE [CASE ...]   CASE = backbone:B:N:same|varied:f32|bf16[:seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "alignnet-3d_amd")]
import alignnet3d  # noqa: E402
from oracle import alignnet_ref as R  # noqa: E402
from tests import test_train_gpu as TT  # noqa: E402
from tests.helpers import oracle_params, varied_pairs  # noqa: E402
from tests.test_fullsize_gpu import _grad_compare  # noqa: E402


def diag(case):
    backbone, B, N, kind, dtype, seed = (case.split(":") + ["5"])[:6]
    B, N, bf16, seed = int(B), int(N), dtype == "bf16", int(seed)
    cfg = alignnet3d.default_model_config()
    cfg["model"]["num_points"], cfg["model"]["backbone"], cfg["training"]["batch_size"] = N, backbone, B
    spec, P32 = oracle_params(cfg, seed=seed)
    d = (varied_pairs if kind == "varied" else R.synth_pairs)(B, N, seed=seed, dtype=np.float32)
    rng = np.random.default_rng(seed)
    du = {k: rng.uniform(size=(B, 256)).astype(np.float32) for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")}
    eng = alignnet3d.Engine(cfg)
    eng.set_variables(P32)
    eng.set_option("train_matmul_bf16", int(bf16))
    res = eng.train_forward_backward(d["pcs1"], d["pcs2"], d, [du[k] for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")])
    dec = eng.debug_train_decisions(B, relu=True)
    ge = {n: eng.get_gradient(n).astype(np.float64) for n in R.trainable_names(spec)}
    decay = eng.state()["bn_decay"]
    ep64, loss64, g64, ema64 = TT._oracle(cfg, P32, d, du, decay, bf16_lift=bf16, checkpoint=True, pinned=dec)
    ep32, loss32, g32, ema32 = TT._oracle(cfg, P32, d, du, decay, bf16_lift=bf16, checkpoint=True, pinned=dec, dt=np.float32)
    emae = {k: eng.get_variable(k) for k in ema64}
    eng.close()
    for tag, g, ep, loss, ema in (("engine", ge, res, res["loss"], emae), ("torch fp32", g32, ep32, loss32, ema32)):
        _, relf, cos, rl2, gs = _grad_compare(lambda n: g[n], spec, g64)
        pred = {k: float(np.abs(np.asarray(ep[k], np.float64) - ep64[k]).max()) for k in ep64}
        print("%s  %-10s vs pinned fp64 oracle: loss %.3e  relative L2 %.3e  cosine %.10f  worst prediction %.2e (%s)" %
              (case, tag, abs(loss - loss64), rl2, cos, max(pred.values()), max(pred, key=pred.get)))
        for n, e in sorted(relf.items(), key=lambda kv: -kv[1])[:8]:
            print("      %-52s %.2e   (own max %.2e of the gradient's %.2e)" % (n, e, float(np.abs(g64[n]).max()), gs))
        # batch statistics through the EMA shadows (s <- d s + (1 - d) batch value): error relative to the layer's largest |value|
        er = {k: float(np.abs(np.asarray(ema[k], np.float64).ravel() - v.ravel()).max() / max(float(np.abs(v).max()), 1e-30)) for k, v in ema64.items()}
        print("      batch statistics (EMA shadows), worst layers: " + ", ".join("%s %.1e" % (k.replace("siamese", "s").replace("/embedding", "/e").replace("transformer", "t").replace("moving_", ""), e)
                                                                              for k, e in sorted(er.items(), key=lambda kv: -kv[1])[:10]))
        # the same per CHANNEL (a BatchNorm divides by its own channel's deviation): worst |error| / |value| over the channels of a layer's variance
        ec = {k: float((np.abs(np.asarray(ema[k], np.float64).ravel() - v.ravel()) / np.maximum(np.abs(v.ravel()), 1e-30)).max()) for k, v in ema64.items() if k.endswith("moving_var")}
        print("      variance per channel, worst layers: " + ", ".join("%s %.1e" % (k.replace("siamese", "s").replace("/embedding", "/e").replace("transformer", "t").replace("/bn/moving_var", ""), e)
                                                                  for k, e in sorted(ec.items(), key=lambda kv: -kv[1])[:10]))
        if os.environ.get("PREDICTIONS"):
            for k in pred:
                print("      prediction %-30s %.2e" % (k, pred[k]))
    sys.stdout.flush()


def same_pins(a, b):
    if isinstance(a, dict):
        return a.keys() == b.keys() and all(same_pins(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(same_pins(x, y) for x, y in zip(a, b))
    return np.array_equal(a, b)


def variants(case, vs):
    """the same step under engine options (kernel variants of the same arithmetic): which one moves the distance from the pinned oracle?"""
    backbone, B, N, kind, dtype, seed = (case.split(":") + ["5"])[:6]
    B, N, bf16, seed = int(B), int(N), dtype == "bf16", int(seed)
    cfg = alignnet3d.default_model_config()
    cfg["model"]["num_points"], cfg["model"]["backbone"], cfg["training"]["batch_size"] = N, backbone, B
    spec, P32 = oracle_params(cfg, seed=seed)
    d = (varied_pairs if kind == "varied" else R.synth_pairs)(B, N, seed=seed, dtype=np.float32)
    rng = np.random.default_rng(seed)
    du = {k: rng.uniform(size=(B, 256)).astype(np.float32) for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")}
    cache = []
    for v in vs:
        eng = alignnet3d.Engine(cfg)
        eng.set_variables(P32)
        eng.set_option("train_matmul_bf16", int(bf16))
        for kv in filter(None, v.split(",")):
            k, val = kv.split("=")
            eng.set_option(k, int(val))
        eng.train_forward_backward(d["pcs1"], d["pcs2"], d, [du[k] for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")])
        dec = eng.debug_train_decisions(B, relu=True)
        if bf16 and backbone == "pointnet":
            dec["round"] = eng.debug_train_rounded(B)   # (bf16: the operand roundings pinned too)
        ge = {n: eng.get_gradient(n).astype(np.float64) for n in R.trainable_names(spec)}
        decay = eng.state()["bn_decay"]
        eng.close()
        g64 = next((g for dd, g in cache if same_pins(dd, dec)), None)
        reused = g64 is not None
        if not reused:
            _, _, g64, _ = TT._oracle(cfg, P32, d, du, decay, bf16_lift=bf16, checkpoint=True, pinned=dec)
            cache.append((dec, g64))
        _, relf, cos, rl2, gs = _grad_compare(lambda n: ge[n], spec, g64)
        print("%s  [%s]%s relative L2 %.3e  worst %s" % (case, v or "default", " (same pins as an earlier variant)" if reused else "", rl2,
                                                       [(n.replace("siamese", "s"), float("%.2g" % e)) for n, e in sorted(relf.items(), key=lambda kv: -kv[1])[:4]]), flush=True)


if __name__ == "__main__":
    if os.environ.get("VARIANTS") is not None:
        for c in sys.argv[1:]:
            variants(c, os.environ["VARIANTS"].split(";"))
    else:
        for c in sys.argv[1:]:
            diag(c)
