"""Shared helpers for the parity tests (the oracle is the checker, never the product)."""
import numpy as np

from oracle import alignnet_ref as R


def small_cfg(N=128, nb=12, s1=(32, 64, 96), s2=(32, 64, 128), emb=(32, 64, 160), fc=(64, 32), backbone="pointnet"):
    return {
        "data": {"num_channels": 3, "ntrain": 1000},
        "model": {
            "backbone": backbone, "num_points": N,
            "options": {
                "angle_factor": 1.0, "early_stage_factor": 0.5,
                "s1transformer": [list(s1), [list(fc), 0.7]],
                "s2transformer": [list(s2), [list(fc), 0.7]],
                "embedding": list(emb),
                "remaining_transform_prediction": [list(fc), 0.7],
            },
            "angles": {"num_bins": nb, "accept_inverted_angle": True},
        },
        "training": {
            "batch_size": 8, "learning_rate": 0.005, "optimizer": {"optimizer": "adam"},
            "lr_extension": {"mode": "decay", "per": "epoch", "step": 30, "rate": 0.5},
            "bn_extension": {"mode": "decay", "per": "epoch", "step": 30, "rate": 0.5, "init": 0.5, "clip": 0.99},
        },
        "gpu_index": 0,
    }


def oracle_params(cfg, seed=0, bn_seed=1):
    """fp32-representable parameters (what the engine holds), as float64 arrays for the oracle."""
    spec = R.NetSpec.from_cfg(cfg)
    P = R.init_params(spec, seed, np.float32)
    R.randomize_bn(P, bn_seed)
    return spec, {k: v.astype(np.float32) for k, v in P.items()}


def logit_margin(logits, nb):
    s = np.sort(logits[:, :nb], axis=1)
    return s[:, -1] - s[:, -2]


def compare_forward(ep_hip, ep_ref, nb, atol=1e-4, rtol=1e-4, margin_eps=1e-3):
    """Bar (BASELINE.json north_star): outputs within 1e-4 in fp32.  The yaw argmax in the
    middle of the network (models/tp8.py:296) is discontinuous: pairs whose top-2 class-logit
    margin is below margin_eps in either tower are excluded from the stage-3 outputs and counted."""
    m1 = logit_margin(ep_ref["pred_pc1angle_logits"], nb)
    m2 = logit_margin(ep_ref["pred_pc2angle_logits"], nb)
    stable = (m1 > margin_eps) & (m2 > margin_eps)
    worst = {}
    for k in ep_ref:
        a, b = ep_hip[k].astype(np.float64), ep_ref[k].astype(np.float64)
        if k in ("pred_translations", "pred_remaining_angle_logits"):
            a, b = a[stable], b[stable]
        err = np.abs(a - b)
        worst[k] = float(err.max()) if err.size else 0.0
        assert np.all(err <= atol + rtol * np.abs(b)), (k, worst[k])
    return worst, int((~stable).sum())


def varied_pairs(B, N, seed=1234, dtype=np.float32, lo=0.4, hi=1.6):
    """synth_pairs with objects of different size and proportions: every pair's box is stretched by its own U(lo, hi) factor per axis in
    the object frame (both clouds of a pair show the same object; labels unchanged).  Why the tests want it: alignnet3d/synth.py draws ONE
    box shape for every pair, so after stage 3's canonicalisation (models/tp8.py:122-127) all clouds of a batch look alike, the pooled
    features differ between samples only by sampling noise, and the heads' batch normalisation (utils/tf_util.py:455-492) divides by that
    small spread: the fp64 oracle's own gradient then moves by 1e-3 .. 1e-2 when its inputs move by one fp32 rounding
    (tools/pinned_report.py prints it).  Real batches hold different objects; with varied boxes the same perturbation moves the
    gradient by ~1e-5 .. 1e-4, and a sharp bar means something."""
    rng = np.random.default_rng(seed + 77)
    d = {k: v.astype(np.float64) for k, v in R.synth_pairs(B, N, seed=seed, dtype=np.float64).items()}
    sc = rng.uniform(lo, hi, (B, 1, 3))
    for k, c, a in (("pcs1", "pc1_centers", "pc1_angles"), ("pcs2", "pc2_centers", "pc2_angles")):
        p = d[k] - d[c][:, None, :]
        cs, sn = np.cos(d[a][:, 0])[:, None], np.sin(d[a][:, 0])[:, None]
        q = np.stack([p[..., 0] * cs + p[..., 1] * sn, -p[..., 0] * sn + p[..., 1] * cs, p[..., 2]], -1) * sc   # object frame, stretched
        d[k] = np.stack([q[..., 0] * cs - q[..., 1] * sn, q[..., 0] * sn + q[..., 1] * cs, q[..., 2]], -1) + d[c][:, None, :]
    return {k: v.astype(dtype) for k, v in d.items()}
