#!/usr/bin/env python3
"""GPU: one training step of the engine against fp64 autograd -- FREE (the oracle decides for itself), PINNED to the engine's
decisions (alignnet_debug_train_decisions), RELU = pinned to the decisions (incl. the loss's angle classes) and to the sign every relu saw (alignnet_debug_train_relu_mask:
the step is then a smooth function of its inputs), ROUND (bf16 cases) = also to the bf16-rounded activations (alignnet_debug_train_rounded), and each pinned oracle against ITSELF with its inputs moved by one fp32 rounding (6e-8
relative): what any fp32 evaluation of this batch can be expected to reproduce.  PINNED_MODES=free,pinned,... selects; RELU_DETAIL=1 lists
the layers whose mask disagrees with the oracle's own signs.  Prints per case: predictions, loss, whole-gradient
cosine / relative L2, the worst tensors, the decision gaps.
Usage: python tools/pinned_report.py CASE [CASE ...]   CASE = backbone:B:N:kind:dtype[:seed], e.g. pointnet:256:1024:varied:f32"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "alignnet-3d_amd")]
import alignnet3d  # noqa: E402
from oracle import alignnet_ref as R  # noqa: E402
from tests import test_train_gpu as TT  # noqa: E402
from tests.helpers import oracle_params, varied_pairs  # noqa: E402


def grad_cmp(ga, gb, spec):
    gs = max(float(np.abs(v).max()) for v in gb.values())
    skip = {(f"siamese/{L.name}" if L.siamese else L.name) + "/biases" for L in R.layer_table(spec) if L.bn}
    names = [n for n in R.trainable_names(spec) if n not in skip]
    rel = {n: float(np.abs(ga[n].reshape(gb[n].shape) - gb[n]).max()) / (float(np.abs(gb[n]).max()) + 1e-5 * gs) for n in names}
    a = np.concatenate([np.asarray(ga[n], np.float64).ravel() for n in names]); b = np.concatenate([np.asarray(gb[n], np.float64).ravel() for n in names])
    return rel, float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))), float(np.linalg.norm(a - b) / np.linalg.norm(b))


def report(case):
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
    dec_nr = {k: v for k, v in dec.items() if k not in ("relu", "loss_cls")}
    dec_round = None
    if bf16 and backbone == "pointnet" and any(m.startswith("round") for m in MODES):
        dec_round = dict(dec, round=eng.debug_train_rounded(B))   # + the bf16-rounded h1 / h2 the step multiplied
    ge = {n: eng.get_gradient(n).astype(np.float64) for n in R.trainable_names(spec)}
    decay = eng.state()["bn_decay"]
    eng.close()
    out = {}
    for mode in MODES:
        t0 = time.time()
        rep = []
        dd = d
        if mode.endswith("+1ulp"):
            dd = {k: v.astype(np.float64) for k, v in d.items()}
            r2 = np.random.default_rng(99)
            for k in ("pcs1", "pcs2"):
                dd[k] = dd[k] * (1 + 6e-8 * r2.standard_normal(dd[k].shape))
        ep, loss, g, _ = TT._oracle(cfg, P32, dd, du, decay, bf16_lift=bf16, checkpoint=True, pinned=None if mode == "free" else (dec_round if mode.startswith("round") and dec_round else dec if mode.startswith(("relu", "round")) else dec_nr), report=rep)
        out[mode] = (ep, loss, g)
        ref = (res, res["loss"], ge) if not mode.endswith("+1ulp") else out[mode[:-5]]
        what = "engine vs oracle" if not mode.endswith("+1ulp") else "oracle vs oracle"
        pred = max(float(np.abs(ref[0][k] - ep[k]).max()) for k in ep)
        rel, cos, rl2 = grad_cmp(ref[2], g, spec)
        top = sorted(rel.items(), key=lambda kv: -kv[1])[:3]
        gaps = {}
        for w, gap, scale, differ, total in (rep[0] if rep else []):
            k = w.split(":")[0]
            if k == "relu" and os.environ.get("RELU_DETAIL") and differ:
                print("    relu %-44s disagree %6d of %10d worst |bn(z)| %.2e scale %.2e" % (w[5:], differ, total, gap, scale))
            gg = gaps.get(k, (0.0, 0, 0)); gaps[k] = (max(gg[0], gap / max(scale, 1.0)), gg[1] + differ, gg[2] + total)
        print("%-34s %-12s %s: pred %.2e loss %.2e | gradient cos %.8f rl2 %.2e worst %s | gaps %s (%.0f s)"
              % (case, mode, what, pred, abs(ref[1] - loss) / max(1.0, abs(loss)), cos, rl2, [(k.replace("siamese", "s"), float("%.2g" % v)) for k, v in top],
                 {k: (float("%.1e" % v[0]), v[1], v[2]) for k, v in gaps.items()}, time.time() - t0), flush=True)


MODES = tuple(os.environ.get("PINNED_MODES", "free,pinned,pinned+1ulp,relu,relu+1ulp").split(","))

if __name__ == "__main__":
    for c in sys.argv[1:]:
        report(c)
