#!/usr/bin/env python3
"""GPU: does bf16 training (BASELINE.json configs[2], option train_matmul_bf16) train like fp32?  One committed A/B.

A fixed synthetic dataset (alignnet3d/synth.py; `--train` training examples + `--held` held-out ones, 1500 points per cloud) is
uploaded to HBM once and every run trains on it through the device sampler (alignnet_train_step_dataset: resample with replacement
to N = 1024 + jitter, provider.py:60-71,97-98) -- the reference's loop train.py:335-383 with batch 256, Adam, the default LR / BN-decay
staircases (30-epoch steps: at 16 steps per epoch the LR halves every 480 steps) -- for `--steps` steps, fp32 and bf16 with `--seeds`
seeds each (seed = Xavier initialisation + dropout stream + batch order).  Every `--every` steps the eval-mode forward (EMA
statistics) runs on the held-out examples and the reference's evaluation metrics are taken exactly as train.py:447-462 /
evaluation.py:128-289 take them: yaw from classLogits2angle (models/tp8.py:229-244, quirk A6(i) kept), pred_angle = a2 - a1 + a_rem,
translation moved to the ground-truth rotation centre, then mean planar translation error, mean angle error (inverted angle accepted,
as the Cars configs do) and the three correctness levels (0.02 / 0.1 / 0.2 m and 1 / 5 / 10 degrees).

Writes the curves as JSON (profiles/r05_convergence.json when run by tools/gpu.sh) and prints a table.  The -m gpu test
tests/test_fullsize_gpu.py::test_bf16_converges_like_fp32 runs a shortened version of the same function and asserts on it.
Usage: python tools/convergence_ab.py [--steps 3000] [--seeds 3] [--out FILE]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "alignnet-3d_amd")]


def class_logits_to_angle(logits, nb):
    """models/tp8.py:229-244 (the host decode of train.py:453-455): class centre + the residual AS STORED (not de-normalised, quirk A6(i))."""
    cls = np.argmax(logits[:, :nb], axis=1)
    ang = cls * (2 * np.pi / float(nb)) + logits[np.arange(len(cls)), nb + cls]
    return np.where(ang > np.pi, ang - 2 * np.pi, ang)


def held_out_metrics(eng, rows, labels, nb, batch=256):
    """eval-mode predictions on dataset rows -> the reference's evaluation numbers (evaluation.py:21-39,128-208)."""
    import evaluation
    outs = {}
    for lo in range(0, len(rows), batch):
        ep = eng.forward_rows(rows[lo:lo + batch], seed=7)   # fixed resampling seed: the same points at every evaluation
        for k, v in ep.items():
            outs.setdefault(k, []).append(v)
    ep = {k: np.concatenate(v) for k, v in outs.items()}
    a1, a2, ar = (class_logits_to_angle(ep[k], nb) for k in ("pred_pc1angle_logits", "pred_pc2angle_logits", "pred_remaining_angle_logits"))
    pred_angles = a2 - a1 + ar                                  # train.py:456
    new_t = evaluation.translate_transform_to_new_center_of_rotation(ep["pred_translations"].astype(np.float64), pred_angles,
                                                                     ep["pred_s2_pc1centers"].astype(np.float64), labels["pc1_centers"].astype(np.float64))
    dt, lt, da, la = [], [], [], []
    for i in range(len(rows)):
        d, l = evaluation.eval_translation(new_t[i], labels["translations"][i]); dt.append(float(d)); lt.append(l)
        d, l = evaluation.eval_angle(float(pred_angles[i]), float(labels["rel_angles"][i, 0]), True); da.append(float(d)); la.append(l)
    lt, la = np.asarray(lt), np.asarray(la)
    return {"mean_dist_translation": float(np.mean(dt)), "mean_dist_angle": float(np.mean(da)),
            "corr_levels_translation": lt.mean(0).round(4).tolist(), "corr_levels_angles": la.mean(0).round(4).tolist(),
            "corr_levels": np.minimum(lt, la).mean(0).round(4).tolist(), "finite": bool(np.isfinite(ep["pred_translations"]).all())}


def run_ab(steps=3000, seeds=3, every=250, n_train=4096, n_held=1024, batch=256, num_points=1024, device=0, log=print):
    import alignnet3d
    from alignnet3d.synth import synth_pairs
    pts = 1500
    n = n_train + n_held
    d = synth_pairs(n, pts, seed=20260930, dtype=np.float32)
    off = np.zeros((n + 1, 2), np.int64); off[1:, 0] = off[1:, 1] = np.arange(1, n + 1) * pts
    lab = np.concatenate([d["translations"], d["rel_angles"], d["pc1_centers"], d["pc2_centers"], d["pc1_angles"], d["pc2_angles"]], 1).astype(np.float32)
    held_rows = np.arange(n_train, n)
    held_lab = {k: d[k][n_train:] for k in ("translations", "rel_angles", "pc1_centers")}
    cfg = alignnet3d.default_model_config()
    cfg["model"]["num_points"] = num_points
    cfg["training"]["batch_size"] = batch
    cfg["data"]["ntrain"] = n_train
    nb = cfg["model"]["angles"]["num_bins"]
    runs = []
    for mode in ("f32", "bf16"):
        for seed in range(seeds):
            eng = alignnet3d.Engine(cfg, device=device, seed=100 + seed)
            eng.set_option("train_matmul_bf16", int(mode == "bf16"))
            eng.upload_dataset(d["pcs1"].reshape(-1, 3), d["pcs2"].reshape(-1, 3), off, lab)
            rng = np.random.default_rng(1000 + seed)
            curve, losses, t0 = [], [], time.perf_counter()
            perm, pos = rng.permutation(n_train), 0
            for k in range(steps):
                if pos + batch > n_train:                     # one permutation per epoch, as train.py:340-346
                    perm, pos = rng.permutation(n_train), 0
                r = eng.train_step_rows(perm[pos:pos + batch], seed=seed * 1000003 + k)
                pos += batch
                losses.append(r["loss"])
                if (k + 1) % every == 0 or k + 1 == steps:
                    m = held_out_metrics(eng, held_rows, held_lab, nb, batch)
                    m.update(step=k + 1, train_loss_mean=float(np.mean(losses[-every:])), learning_rate=r["learning_rate"], bn_decay=r["bn_decay"])
                    curve.append(m)
            sec = time.perf_counter() - t0
            eng.close()
            runs.append({"dtype": mode, "seed": seed, "seconds": round(sec, 1), "finite_losses": bool(np.all(np.isfinite(losses))), "curve": curve})
            f = curve[-1]
            log("%-4s seed %d: %d steps in %5.1f s | held-out: translation %.4f m, angle %.2f deg, levels %s | train loss %.4f"
                % (mode, seed, steps, sec, f["mean_dist_translation"], f["mean_dist_angle"], f["corr_levels"], f["train_loss_mean"]))
    return {"what": "fp32 vs bf16-conv training (train_matmul_bf16) on one fixed synthetic dataset through the device sampler; held-out eval-mode metrics "
                    "as evaluation.py takes them", "steps": steps, "seeds": seeds, "every": every, "n_train": n_train, "n_held": n_held, "batch": batch,
            "num_points": num_points, "runs": runs, "summary": summarise(runs)}


def summarise(runs):
    """final held-out metrics per dtype over the seeds (mean, min, max) and bf16's distance from fp32's spread"""
    out = {}
    for key in ("mean_dist_translation", "mean_dist_angle"):
        per = {m: [r["curve"][-1][key] for r in runs if r["dtype"] == m] for m in ("f32", "bf16")}
        out[key] = {m: {"mean": float(np.mean(v)), "min": float(np.min(v)), "max": float(np.max(v))} for m, v in per.items()}
        out[key]["bf16_mean_over_f32_max"] = out[key]["bf16"]["mean"] / max(out[key]["f32"]["max"], 1e-12)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--every", type=int, default=250)
    ap.add_argument("--train", type=int, default=4096)
    ap.add_argument("--held", type=int, default=1024)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    res = run_ab(a.steps, a.seeds, a.every, a.train, a.held)
    print(json.dumps(res["summary"], indent=1))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)
