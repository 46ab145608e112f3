#!/usr/bin/env python3
"""Headline benchmark: AlignNet-3D point-cloud pairs/sec at N=1024 on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode infer|train] [--batch B] [--workload pointnet|dgcnn]

A "step" is one pass of the hot path (models/tp8.py get_model, eval mode, i.e. the reference's timed `sess.run`,
train.py:447-449) over one batch of synthetic pairs that is already resident in HBM.  Workload at N=1 =
BASELINE.json configs[1]: SynthCars widths, batch 256, N=1024, fp32.  With --gpus N (launched under
torch.distributed.run, one rank per GPU) every rank processes its own batch of 256 pairs (weak scaling; pairs are
independent in eval mode so the inference path has no data-path collective); value = all pairs / max-over-ranks time.
For N > 1 the training leg runs by default as well: data-parallel steps with the library's RCCL all-reduce over xGMI.

Prints ONE JSON line (rank 0).  Every timed leg (headline, `infer_bf16x3`, `train`, `train.bf16`) carries a `roofline`
object: dominant kernel = the timed kernel with the largest HIP-event time inside that leg's timed region (events on the
engine's own stream, alignnet_profile_read_kernel); achieved = the FLOPs that kernel's algorithm executes per step
(`kernel_macs`) / its event time; peak = 157.3 TFLOP/s fp32 MFMA or 2500 TFLOP/s dense bf16 MFMA
(MI355X_MICROARCH.md); traffic = HBM bytes of that kernel per launch (traffic_per_step: per step) from the committed rocprofv3 PMC
summary of the same command (profiles/r*_<leg>_pmc_traffic.json, separate --pmc passes, FETCH_SIZE doubled as the guide prescribes;
taken at the batch size of tools/refresh_profiles.sh's line for that leg), else null.
  pcie_inclusive  the reference's own methodology (train.py:447-449 times sess.run including the feed copy): pageable
                  host buffers in, host buffers out, blocking alignnet_forward.  Never the headline `value`.
  cpu_baseline    the oracle ("port": unfused op-by-op NumPy fp32 restatement, eval mode, batch 32 as the reference's
                  timing mode train.py:557) on this box's host cores with 1 / 16 / all BLAS threads; value = the best.
"""
import argparse
import glob
import json
import math
import os
import re
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "alignnet-3d_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

# algorithmic work (SURVEY.md 8d / BASELINE.md 2), SynthCars widths, N=1024, nb=50
FLOPS_PER_PAIR_TOTAL = 1_047_688_704
FLOPS_PER_PAIR_TRAIN_SURVEY = 3.14e9   # dense forward + dX + dW, the count a layer-by-layer backward would execute
N_POINTS = 1024
PEAK_F32, PEAK_BF16 = 157.3, 2500.0    # TFLOP/s, dense (MI355X_MICROARCH.md)
# Vector-instruction issue roofline (MI355X_MICROARCH.md, "Wave scheduling": a wave issues each VALU instruction over 2 cycles on its
# SIMD-32; 256 CUs x 4 SIMDs at 2.4 GHz): the bound of a kernel whose time goes into vector instructions rather than into a matrix pipe
PEAK_VALU_GINSTR = 256 * 4 * 2.4 / 2.0   # = 1228.8 G wave-level VALU instructions / s
# Static vector-instruction counts per wave and unit of work for the kernels DESIGN.md identifies as VALU-issue-bound, used when no
# committed PMC pass (SQ_INSTS_VALU, profiles/r*_pmc_by_kernel.json) of the same command and shape is available:
#   train_bwd_b2 (bf16 mode): ~1.1 k vector instructions per wave and 64-point tile, 4 waves per tile
#   knn: ~700 vector instructions per query point, one wave per query
VALU_INSTR_STATIC = {"train_bwd_b2": lambda n: 3 * ((n + 63) // 64) * 4 * 1100.0, "knn": lambda n: n * 700.0}   # per cloud (three stages / one graph)
K_NEIGHBOURS = 20                      # models/tp8.py:33


def stage_widths(cfg):
    o = cfg["model"]["options"]
    return [o["s1transformer"][0], o["s2transformer"][0], o["embedding"]]


def backbone_macs_per_cloud(cfg):
    """MACs of the three fused backbones for one cloud (the dominant inference kernel's algorithmic work).
    dgcnn (models/tp8.py:30-46): edge convs widths[:-1] on k = 20 edges per point, then widths[-1] per point."""
    n = cfg["model"]["num_points"]
    dg = cfg["model"]["backbone"] == "dgcnn"
    tot = 0
    for widths in stage_widths(cfg):
        if dg:
            cin, edge = 6, 0
            for c in widths[:-1]:
                edge += cin * c
                cin = c
            tot += K_NEIGHBOURS * edge + cin * widths[-1]
        else:
            cin = 3
            for c in widths:
                tot += cin * c
                cin = c
    return tot * n


def _blocks(c):
    """32 x 32 upper-triangle blocks of a c x c Gram matrix (what the Gram kernels execute)."""
    t = (c + 31) // 32
    return t * (t + 1) // 2 * 1024


P3_FUSED_GRAM = True   # fp32 phase 3 of the shipped widths accumulates Gram(h2) in the same pass (set from the engine's options in main())


def kernel_macs(cfg, kernel, bf16, with_gram=True):
    """MACs per CLOUD, summed over the three stages, that `kernel`'s algorithm executes in this design:
    the recompute of the cheap early layers is part of each pass, the dense backward through the C2 -> C3 lift is replaced by
    the Gram identities.  3-layer backbones [C1, C2, C3] (the only trainable shape)."""
    n = cfg["model"]["num_points"]
    dg = cfg["model"]["backbone"] == "dgcnn"
    if kernel == "backbone":
        return backbone_macs_per_cloud(cfg)
    if kernel == "knn":
        return n * n * 3                       # distance products, one graph per cloud (VALU work, not MFMA)
    k0 = 6 if dg else 3
    rows = n * K_NEIGHBOURS if dg else n       # edge rows per cloud in the DGCNN branch
    tot = 0
    for (c1, c2, c3) in [w[:3] for w in stage_widths(cfg)]:
        if kernel == "train_fwd_phase3":       # h1, h2 recomputed from xyz, z3 = h2 W3 (+ the Gram of h2 inside the bf16 kernel)
            # (+ the Gram of h2: inside the bf16 kernels, and inside the fp32 128-point kernel of the shipped widths -- csrc/kernels_train_fwd_wide.h)
            fused_gram = with_gram and (bf16 or ((c1, c2) == (64, 128) and P3_FUSED_GRAM))
            tot += (n * (c2 * c3) if dg else n * (k0 * c1 + c1 * c2 + c2 * c3)) + (n * _blocks(c2) if fused_gram else 0)
        elif kernel == "train_fwd_phase2":     # fp32: h1 + Gram(h1) (statistics of z2 from the Gram); bf16: h1 + z2 = h1 W2
            tot += n * (k0 * c1 + (c1 * c2 if bf16 else _blocks(c1)))
        elif kernel == "train_gram_h2":
            tot += n * _blocks(c2)
        elif kernel == "train_bwd_b2":         # h1, h2 recomputed (pointnet), dh2 = h2 Q3
            tot += n * ((0 if dg else k0 * c1 + c1 * c2) + c2 * c2)
        elif kernel == "train_bwd_b1":         # h1 recomputed, dh1 = dy2 V2 + h1 Q2, U2 = h1^T dy2, Pdy = x'^T dy1
            tot += n * (k0 * c1 + c2 * c1 + c1 * c1 + c1 * c2 + 4 * c1)
        elif kernel == "dg_train_fwd":         # per edge row: K=6 lift, z2 = h1 W2, Gram(h1)
            tot += rows * (k0 * c1 + c1 * c2 + _blocks(c1))
        elif kernel == "dg_train_bwd_edge":    # per edge row: lift, h1 Q2 dense, the two sparse products at 1/k density, Pdy chain
            tot += rows * (k0 * c1 + c1 * c1 + 2 * c1 * c2 // K_NEIGHBOURS + 8 * c1)
    return tot


KERNEL_IN_PROFILE = {   # bench kernel key -> substring of the kernel name in profiles/*_pmc_by_kernel.json
    "train_fwd_phase3": "train_fwd_phase3_wide|train_fwd_phase23<3", "train_fwd_phase2": "train_fwd_gram1|train_fwd_phase23<2", "train_gram_h2": "gram_h2_kernel", "train_bwd_b2": "train_bwd_b2",
    "train_bwd_b1": "train_bwd_b1", "dg_train_fwd": "dg_train_fwd", "dg_train_bwd_edge": "dg_train_bwd_edge", "knn": "knn_kernel"}


def pmc_traffic(leg_tag, kernel_substr, shape):
    """HBM bytes per step of one kernel from the newest committed rocprofv3 PMC summary of this leg, or None.  `shape` =
    (pairs per GPU, points per cloud) of the leg being reported: a summary taken at another shape (or one that does not say at
    which -- rounds 1 / 2) is not quoted."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%spmc_traffic.json" % (leg_tag + "_" if leg_tag else ""))),
                   key=lambda f: int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)))
    files = [f for f in files if re.fullmatch(r"r\d+_%spmc_traffic\.json" % (leg_tag + "_" if leg_tag else ""), os.path.basename(f))]
    if not files:
        return None
    try:
        j = json.load(open(files[-1]))
        sh = j.get("shape") or {}
        if (sh.get("pairs_per_gpu"), sh.get("num_points")) != tuple(shape):
            return None
        for name, v in (j.get("kernels") or {}).items():
            if any(alt in name for alt in kernel_substr.split("|")):
                return {"bytes_per_step": v["hbm_bytes_per_step"], "bytes_per_launch": v["hbm_bytes_per_launch"], "source": os.path.basename(files[-1]),
                        "commit": j.get("commit")}
    except Exception:
        return None
    return None


def pmc_valu_instr(leg_tag, kernel_substr, shape):
    """Wave-level VALU instructions per launch of one kernel (SQ_INSTS_VALU minus the MFMAs, which issue on the matrix pipe) from the
    newest committed PMC summary of this leg at this shape, or None."""
    pre = leg_tag + "_" if leg_tag else ""
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_%spmc_by_kernel.json" % pre)) if re.fullmatch(r"r\d+_%spmc_by_kernel\.json" % pre, os.path.basename(f))]
    files.sort(key=lambda f: int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)))
    if not files:
        return None
    try:
        tf = files[-1].replace("pmc_by_kernel", "pmc_traffic")
        sh = (json.load(open(tf)).get("shape") or {}) if os.path.exists(tf) else {}
        if (sh.get("pairs_per_gpu"), sh.get("num_points")) != tuple(shape):
            return None
        for name, v in json.load(open(files[-1])).items():
            if any(alt in name for alt in kernel_substr.split("|")) and "SQ_INSTS_VALU" in v:
                c = v["SQ_INSTS_VALU"]
                mf = v.get("SQ_INSTS_MFMA")
                per = c["sum"] / max(c["dispatches"], 1)
                if mf:
                    per -= mf["sum"] / max(mf["dispatches"], 1)
                return {"instr_per_launch": per, "source": os.path.basename(files[-1])}
    except Exception:
        return None
    return None


def cpu_baseline(cfg, seconds_per_setting=4.0):
    from oracle import alignnet_ref as R
    from threadpoolctl import threadpool_limits, threadpool_info
    allc = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    spec = R.NetSpec.from_cfg(cfg)
    P = R.init_params(spec, 0, np.float32)
    for k in P:  # plausible eval-mode shadows: mean 0, var 1
        if k.endswith("moving_var"):
            P[k] = np.ones_like(P[k])
    B = 32  # the reference's own timing mode uses bs=32 (train.py:557)
    d = R.synth_pairs(B, spec.num_points, seed=1234, dtype=np.float32)
    sweep = []
    for threads in sorted({1, min(16, allc), allc}):
        with threadpool_limits(limits=threads):
            R.get_model(P, spec, d["pcs1"], d["pcs2"])  # warm-up
            n, t0 = 0, time.perf_counter()
            while True:
                R.get_model(P, spec, d["pcs1"], d["pcs2"])
                n += 1
                dt = time.perf_counter() - t0
                if dt >= seconds_per_setting or n >= 200:
                    break
        sweep.append({"threads": threads, "value": round(B * n / dt, 2), "batches": n, "seconds": round(dt, 1)})
    best = max(sweep, key=lambda s: s["value"])
    return {"value": best["value"], "unit": "pairs/s", "cores": best["threads"], "kind": "port", "host_cores": os.cpu_count(),
            "sweep": sweep,
            "sample": f"batches of {B} pairs (N={spec.num_points}, SynthCars widths, eval mode), unfused NumPy fp32 oracle, "
                      f"~{seconds_per_setting:.0f} s per thread setting, forward only (cf. reference train.py:447-449); value = best setting"}


ALLOWED_ENV = {"ALIGNNET_HIP_LIB"}   # another build of the same sources for same-box A/B runs; reported in the line as `options.library`


def refuse_stray_environment():
    """The library reads no environment variable (tests/test_capi_cpu.py), and a benchmark line must not depend on one either: any
    ALIGNNET_* variable other than the documented switch above aborts the run (round 3: ALIGNNET_DBG skipped work inside the kernels)."""
    stray = sorted(k for k in os.environ if k.startswith("ALIGNNET_") and k not in ALLOWED_ENV)
    if stray:
        sys.stderr.write("bench.py: refusing to run with %s set (kernel variants are engine options `ab_*`, reported in the line)\n" % ", ".join(stray))
        raise SystemExit(2)


def engine_options(eng):
    """The option set the timed engine ran with (every key that changes which kernels run), for the JSON line."""
    import alignnet3d
    keys = ("train_matmul_bf16", "infer_matmul_bf16x3", "train_fused_tail", "train_phase3_tile64", "train_dw_side_stream", "allreduce_overlap",
            "sync_bn", "global_loss", "grad_communicator", "ab_mask", "ab_tiles_per_wg")
    o = {k: eng.get_option(k) for k in keys}
    o["library"] = os.path.basename(alignnet3d.library_path())
    try:   # the ablation build (csrc/ablate.h) knows this key; the shipped library does not
        eng.set_option("ablate_dbg", 0)
        o["ablation_build"] = True
    except Exception:   # noqa: BLE001
        o["ablation_build"] = False
    return o


def extra_legs(eng, local_rank, min_seconds):
    """Short secondary legs appended to the default one-GPU inference run so that the driver's record carries BASELINE.json configs[4]
    (DGCNN, N = 4096) and the SURVEY 8(f) rows: dgcnn inference at one GPU's share of configs[4] (512 pairs), dgcnn training steps
    (fp32 / bf16 convs, 64 pairs), the batch loader (reference-style files vs packed cache on the host; HBM-resident dataset + device
    sampler feeding training steps) and the ICP refinement.  About 6 s in total; never part of `value`."""
    import tempfile
    import torch
    import alignnet3d
    from alignnet3d.synth import synth_pairs
    dev = torch.device("cuda", local_rank)
    out = {}
    t_begin = time.perf_counter()

    def timed(step, sync, floor=3):
        step(); sync()
        t0 = time.perf_counter(); step(); sync()
        est = max(time.perf_counter() - t0, 1e-5)
        k = max(floor, int(math.ceil(min_seconds / est)))
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        sync()
        return (time.perf_counter() - t0) / k, k

    # ---- DGCNN branch (models/tp8.py:30-46), BASELINE.json configs[4]: N = 4096, 4096 pairs over 8 GPUs = 512 per GPU
    try:
        cfg = alignnet3d.default_model_config()
        cfg["model"]["num_points"], cfg["model"]["backbone"] = 4096, "dgcnn"
        Bi, Bt = 512, 64
        cfg["training"]["batch_size"] = Bt
        dg = alignnet3d.Engine(cfg, device=local_rank, seed=0)
        for name, shp, _ in dg.variables():
            if name.endswith("moving_var"):
                dg.set_variable(name, np.ones(shp[0] * shp[1], np.float32))
        d = synth_pairs(Bi, 4096, seed=4321, dtype=np.float32)
        p1, p2 = torch.from_numpy(d["pcs1"]).to(dev), torch.from_numpy(d["pcs2"]).to(dev)
        lab = {k: torch.from_numpy(np.ascontiguousarray(d[k][:Bt])).to(dev) for k in ("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles")}
        nb2 = 2 * cfg["model"]["angles"]["num_bins"]
        outs = {k: torch.empty(Bi, nb2 if "logits" in k else 3, device=dev) for k in alignnet3d.OUTPUT_NAMES}
        ptrs = {k: v.data_ptr() for k, v in outs.items()}
        dg.profile_enable(True); dg.profile_read(reset=True)
        sec, k = timed(lambda: dg.forward_device(p1.data_ptr(), p2.data_ptr(), Bi, ptrs), dg.synchronize, floor=2)
        kern = dg.profile_kernels(); dg.profile_read(reset=True)
        n_timed = k + 2
        bb_ms = kern.get("backbone", (0.0, 0))[0] / n_timed
        flops = 2.0 * backbone_macs_per_cloud(cfg) * 2 * Bi
        out["dgcnn"] = {"infer": {"value": round(Bi / sec, 1), "unit": "pairs/s", "ms_per_step": round(sec * 1e3, 3), "steps": k, "pairs_per_step": Bi, "num_points": 4096,
                                  "dtype": "f32", "kernel": dg.last_backbone_kernel(), "backbone_ms_per_step": round(bb_ms, 3),
                                  "knn_ms_per_step": round(kern.get("knn", (0.0, 0))[0] / n_timed, 3),
                                  "roofline_frac": round(flops / (bb_ms * 1e-3) / 1e12 / PEAK_F32, 4) if bb_ms > 0 else None,
                                  "what": "BASELINE.json configs[4] at one GPU's share (512 of 4096 pairs), eval-mode forward, inputs in HBM"}}
        labp = {k: v.data_ptr() for k, v in lab.items()}
        for tdtype in ("f32", "bf16"):
            dg.set_option("train_matmul_bf16", int(tdtype == "bf16"))
            dg.profile_read(reset=True)
            sec, k = timed(lambda: dg.train_step_device(p1.data_ptr(), p2.data_ptr(), labp, Bt), dg.synchronize, floor=2)
            kern = dg.profile_kernels(); dg.profile_read(reset=True)
            out["dgcnn"]["train_" + tdtype] = {"value": round(Bt / sec, 1), "unit": "pairs/s", "ms_per_step": round(sec * 1e3, 3), "steps": k, "pairs_per_step": Bt,
                                               "num_points": 4096, "kernel_ms_per_step": {n: round(v[0] / (k + 2), 3) for n, v in sorted(kern.items(), key=lambda kv: -kv[1][0])[:5]},
                                               "last_train_kernel": dg.get_option("last_train_kernel")}
        dg.profile_enable(False)
        dg.close()
        del p1, p2, outs
    except Exception as e:   # noqa: BLE001 -- a secondary leg must not cost the line
        out["dgcnn"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- batch loader: provider.load_batch on reference-style files vs the packed cache (host), then the HBM-resident dataset
    n_ex, pts = 256, 1500
    dd = synth_pairs(n_ex, pts, seed=99, dtype=np.float32)
    try:
        import config as cfgmod
        import provider
        with tempfile.TemporaryDirectory() as tmp:
            root = os.path.join(tmp, "SynthBench")
            for sub in ("meta", "pointcloud1", "pointcloud2", "split"):
                os.makedirs(os.path.join(root, sub))
            txt = lambda v: "\n".join("%.18e" % x for x in np.ravel(v)) + "\n"
            for i in range(n_ex):
                json.dump({"translation": txt(dd["translations"][i]), "rel_angle": float(dd["rel_angles"][i, 0]), "start_position": txt(dd["pc1_centers"][i]),
                           "end_position": txt(dd["pc2_centers"][i]), "start_angle": float(dd["pc1_angles"][i, 0]), "end_angle": float(dd["pc2_angles"][i, 0])},
                          open(os.path.join(root, "meta", "%08d.json" % i), "w"))
                np.save(os.path.join(root, "pointcloud1", "%08d.npy" % i), dd["pcs1"][i]); np.save(os.path.join(root, "pointcloud2", "%08d.npy" % i), dd["pcs2"][i])
            for f in ("train", "val"):
                open(os.path.join(root, "split", f + ".txt"), "w").write("\n".join(map(str, range(n_ex))) + "\n")
            cfgp = os.path.join(tmp, "c.json")
            json.dump({"data": {"basepath": root}, "logging": {"basedir": tmp}, "model": {"num_points": 1024}, "training": {"batch_size": 128}}, open(cfgp, "w"))
            cfgmod.reset_config()
            cfgmod.load_config(cfgp)
            idx = list(range(128))
            np.random.seed(0); t = time.perf_counter(); a = provider.load_batch(idx); t_file = time.perf_counter() - t
            t = time.perf_counter(); provider.use_packed_cache(); t_pack = time.perf_counter() - t
            np.random.seed(0); t = time.perf_counter(); b = provider.load_batch(idx); t_packed = time.perf_counter() - t
            same = all(np.array_equal(x, y) for x, y in zip(a, b))
            cfgmod.reset_config()
        out["loader"] = {"file_based_pairs_per_s": round(128 / t_file, 1), "packed_pairs_per_s": round(128 / t_packed, 1), "packing_seconds": round(t_pack, 3),
                         "identical_batches": bool(same), "examples": n_ex, "points_per_cloud": pts,
                         "what": "provider.load_batch (provider.py:85-136) of 128 examples resampled to N = 1024: per-example json + npy files as the reference reads them, "
                                 "vs the packed cache (alignnet3d/packed.py), same seeded batch"}
    except Exception as e:   # noqa: BLE001
        out["loader"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- HBM-resident dataset + device sampler feeding training steps, and the ICP refinement on the same uploaded clouds
    try:
        import evaluation
        off = np.zeros((n_ex + 1, 2), np.int64); off[1:, 0] = off[1:, 1] = np.arange(1, n_ex + 1) * pts
        labt = np.concatenate([dd["translations"], dd["rel_angles"], dd["pc1_centers"], dd["pc2_centers"], dd["pc1_angles"], dd["pc2_angles"]], 1).astype(np.float32)
        eng.upload_dataset(dd["pcs1"].reshape(-1, 3), dd["pcs2"].reshape(-1, 3), off, labt)
        rng = np.random.default_rng(0)
        inits = [evaluation.get_mat_angle(dd["translations"][i] + rng.normal(0, 0.05, 3), float(dd["rel_angles"][i, 0]) + rng.normal(0, 0.03), rotation_center=dd["pc1_centers"][i])
                 for i in range(n_ex)]
        rows = np.arange(n_ex)
        eng.icp_refine_rows(rows, inits, 0.1, 30)
        t = time.perf_counter(); res = eng.icp_refine_rows(rows, inits, 0.1, 30); dt = time.perf_counter() - t
        out["icp"] = {"value": round(n_ex / dt, 1), "unit": "pairs/s", "pairs": n_ex, "points_per_cloud": pts, "ms": round(dt * 1e3, 2), "radius": 0.1, "max_iterations": 30,
                      "mean_iterations": round(float(res["iterations"].mean()), 2), "mean_fitness": round(float(res["fitness"].mean()), 3),
                      "what": "alignnet_icp_refine_dataset (icp.py:69-78 / train.py:463-484: point-to-point, rotation about z), seeded near the truth like the network's prediction"}
        k = [0]
        def srow():
            eng.train_step_rows(rng.integers(0, n_ex, 256), seed=k[0]); k[0] += 1
        sec, ks = timed(srow, eng.synchronize, floor=5)
        out["loader"]["device_sampler_train"] = {"value": round(256 / sec, 1), "unit": "pairs/s", "ms_per_step": round(sec * 1e3, 3), "steps": ks,
                                                 "what": "alignnet_train_step_dataset: batch of 256 drawn on the device from the HBM-resident dataset (resample with replacement + jitter, "
                                                         "provider.py:60-71,97-98) + the fp32 training step, no host batch"}
    except Exception as e:   # noqa: BLE001
        out["icp"] = out.get("icp") or {"error": "%s: %s" % (type(e).__name__, e)}
    out["seconds"] = round(time.perf_counter() - t_begin, 2)
    return out


def visible_gpus():
    import torch
    return torch.cuda.device_count()


def self_launch(n):
    """Re-execute this command under torch.distributed.run with one rank per GPU (RCCL over xGMI inside the library).
    Refuses -- non-zero exit, message on stderr -- when the node shows fewer than n GPUs: ranks never share a device."""
    import socket
    import subprocess
    ndev = visible_gpus()
    if ndev < n:
        sys.stderr.write(f"bench.py: --gpus {n} but only {ndev} GPU(s) visible: refusing to share devices (one rank per GPU)\n")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def sclk_mhz(device):
    """Current shader clock of one GPU from rocm-smi (None when it cannot be read)."""
    import subprocess
    try:
        out = subprocess.run(["rocm-smi", "-d", str(device), "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        j = json.loads(out[out.index("{"):])
        for card in j.values():
            for k, v in card.items():
                if "sclk" in k.lower() and "speed" in k.lower():    # {"card0": {"sclk clock speed:": "(2400Mhz)", ...}}
                    m = re.search(r"(\d+)\s*Mhz", str(v), re.I)
                    if m:
                        return int(m.group(1))
    except Exception:
        return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="pairs per GPU per step")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer")
    ap.add_argument("--train-dtype", choices=["f32", "bf16"], default="f32",
                    help="--mode train: operand type of the dominant training GEMMs (bf16 = BASELINE.json configs[2])")
    ap.add_argument("--infer-dtype", choices=["f32", "bf16x3"], default="f32",
                    help="inference backbone arithmetic of the timed region: exact fp32 MFMA (default, BASELINE.json configs[1]) or the "
                         "opt-in split-bf16 mode (three bf16 MFMAs per fp32 product, fp32 accumulate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-leg", action="store_true")
    ap.add_argument("--no-split-leg", action="store_true")
    ap.add_argument("--no-pcie-leg", action="store_true")
    ap.add_argument("--train-leg", action="store_true", help="(kept for compatibility: the training leg now runs by default, also for --gpus > 1)")
    ap.add_argument("--allreduce-overlap", type=int, default=1, help="data-parallel training: bucketed all-reduce next to the backward (1) or one call after it (0)")
    ap.add_argument("--workload", choices=["pointnet", "dgcnn"], default="pointnet",
                    help="dgcnn = BASELINE.json configs[4] shape: N=4096, edge-conv branch (--mode train: fp32 only)")
    ap.add_argument("--points", type=int, default=0, help="points per cloud (default 1024; 4096 for dgcnn)")
    ap.add_argument("--min-leg-seconds", type=float, default=0.35, help="secondary legs are timed for at least this long")
    ap.add_argument("--sync-bn", type=int, default=0, help="data-parallel training legs (--gpus > 1): 1 = BatchNorm statistics and the loss over the global batch "
                                                             "(engine options sync_bn + global_loss: the reference's single-device semantics); 0 = local BN / local loss")
    ap.add_argument("--sustained-seconds", type=float, default=5.0,
                    help="after the K timed steps: the same step back to back for at least this long (clock-settled rate + sclk readings); 0 = off")
    ap.add_argument("--secondary-timeout", type=float, default=240.0,
                    help="--gpus > 1 under torch.distributed.run: if the legs AFTER the headline measurement (single-rank reference, sustained loop, training legs over "
                         "RCCL) have not finished after this many seconds, rank 0 prints the line with what is complete and every rank exits (0 = no watchdog)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the short dgcnn / loader / icp legs the default --gpus 1 inference run appends")
    ap.add_argument("--grad-communicator", type=int, default=0,
                    help="data-parallel training: 1 = the gradient buckets travel on a second RCCL communicator of their own (alignnet_comm_init_grad), so that "
                         "sync-BN's per-layer sums do not serialise behind them; 0 = one communicator for everything")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal of the multi-GPU code path on ONE GPU: re-launch under torch.distributed.run with one rank and take every branch a "
                         "--gpus N > 1 run takes (NCCL process group with device_id, the library's RCCL communicator, barriers, max over ranks, the "
                         "sustained loop's flag all-reduce, per-rank spread); the line says so in `forced_dist`")
    ap.add_argument("--rehearse-world", type=int, default=0,
                    help="rehearsal of the --gpus W > 1 code path on ONE GPU: W ranks as host threads of this process, each with its own engine on device 0, the "
                         "library's in-process loopback communicator standing where RCCL stands and a thread rendezvous where torch.distributed stands.  Every "
                         "world > 1 branch runs -- collectives that must be entered by EVERY rank, the per-rank gather, rccl_ranks, the exposed all-reduce -- so "
                         "that the driver's multi-GPU node is not the first to run them.  NOT a measurement: W ranks share one device (the line says so)")
    args = ap.parse_args()
    refuse_stray_environment()

    if args.rehearse_world > 1:
        raise SystemExit(rehearse(args))
    if (args.gpus > 1 or args.force_dist) and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: spawn the N ranks (one per GPU) here instead of silently running one
        raise SystemExit(self_launch(args.gpus))
    # The contract is ONE line on stdout.  Libraries write there too (librccl prints a version banner on its first communicator -- found by the
    # world-1 rehearsal, round 5): everything this process and its libraries print goes to stderr, the JSON line alone to the real stdout.
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    run_rank(args, real_stdout, None)


class ThreadRanks:
    """What bench.py uses of torch.distributed, for W ranks that are THREADS of one process (--rehearse-world): one instance per rank over a shared
    rendezvous.  A collective that not every rank enters does not complete: the barrier times out and the run fails (the defect class of
    round 5's rank-0-only sync-BN probe)."""

    class ReduceOp:
        SUM, MAX = "sum", "max"

    class Shared:
        def __init__(self, world, timeout):
            import threading
            self.world, self.timeout = world, timeout
            self.barrier = threading.Barrier(world, timeout=timeout)
            self.slots = [None] * world

    def __init__(self, rank, shared):
        self.rank, self.shared = rank, shared

    def get_rank(self): return self.rank
    def get_world_size(self): return self.shared.world
    def get_backend(self): return "loopback-threads"
    def is_initialized(self): return True
    def barrier(self): self.shared.barrier.wait()
    def destroy_process_group(self): pass

    def _exchange(self, value):
        sh = self.shared
        sh.barrier.wait()              # (the previous collective's readers are done with the slots)
        sh.slots[self.rank] = value
        sh.barrier.wait()
        return list(sh.slots)

    def all_reduce(self, t, op="sum"):
        import torch
        vals = self._exchange(t.detach().clone())
        st = torch.stack(vals)
        t.copy_(st.max(0).values if op == "max" else st.sum(0))

    def all_gather(self, out, t):
        vals = self._exchange(t.detach().clone())
        for o, v in zip(out, vals):
            o.copy_(v)

    def broadcast_object_list(self, box, src=0):
        vals = self._exchange(box[0])
        box[0] = vals[src]


def rehearse(args):
    """--rehearse-world W: run_rank on W threads (one engine each, device 0), joined by ThreadRanks + the loopback communicator."""
    import threading
    W = args.rehearse_world
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    shared = ThreadRanks.Shared(W, timeout=float(os.environ.get("BENCH_REHEARSAL_TIMEOUT_S", "180")))
    errors = []
    import torch
    import alignnet3d
    if torch.cuda.device_count() < 1:   # (the device runtime and the library are initialised here, once, not by W threads at the same time)
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.init()
    torch.zeros(1, device="cuda:0")
    alignnet3d.load_library()

    def body(r):
        try:
            run_rank(args, real_stdout, ThreadRanks(r, shared))
        except BaseException as e:   # noqa: BLE001 -- a failed rank must take the others' barriers down with it
            errors.append((r, "%s: %s" % (type(e).__name__, e)))
            shared.barrier.abort()
    threads = [threading.Thread(target=body, args=(r,)) for r in range(W)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for r, e in sorted(errors):
        sys.stderr.write("bench.py --rehearse-world: rank %d failed: %s\n" % (r, e))
    return 1 if errors else 0


def run_rank(args, real_stdout, threads):
    """One rank of the bench: a process under torch.distributed.run (threads is None) or a thread of --rehearse-world (threads: its ThreadRanks)."""
    import torch
    import alignnet3d
    from alignnet3d.synth import synth_pairs  # synthetic input generator (SURVEY 8d recipe); the oracle is only imported by cpu_baseline()

    if threads is not None:
        rank, local_rank, world = threads.get_rank(), 0, threads.get_world_size()
        args.gpus = world                         # (what the line reports as ranks; devices_used says one)
        args.no_cpu_baseline = args.no_pcie_leg = args.no_extra_legs = True
        args.sustained_seconds = min(args.sustained_seconds, 0.5)
    else:
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist_on = world > 1 or args.force_dist   # every `world > 1` branch below; --force-dist takes them at world = 1
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    local_world = 1 if threads is not None else int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if local_world > ndev:
        # one rank per GPU is the contract; sharing a device would report n_gpus ranks' worth of throughput from fewer GPUs
        raise SystemExit(f"{local_world} ranks on this node but only {ndev} GPU(s) visible: refusing to share devices")
    dist = None
    if threads is not None:
        dist = threads
        torch.cuda.set_device(0)
    elif dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cfg = alignnet3d.default_model_config()
    npts = args.points or (4096 if args.workload == "dgcnn" else N_POINTS)
    cfg["model"]["num_points"] = npts
    cfg["model"]["backbone"] = args.workload
    dg = args.workload == "dgcnn"
    if dg:
        args.no_train_leg = args.no_cpu_baseline = args.no_split_leg = args.no_pcie_leg = True
    cfg["training"]["batch_size"] = args.batch * world
    B = args.batch
    eng = alignnet3d.Engine(cfg, device=local_rank, seed=0)
    opts0 = engine_options(eng)
    if opts0["ablation_build"]:
        raise SystemExit("bench.py: %s is the ablation build (result-changing timing switches compiled in): not benchmarked" % opts0["library"])
    global P3_FUSED_GRAM
    P3_FUSED_GRAM = not opts0["train_phase3_tile64"] and not (opts0["ab_mask"] & (1 << 6))   # csrc/engine.h: AB_P3_NOGRAM
    # eval-mode shadows: mean 0 / var 1 (a freshly initialised net has var 0 -> degenerate scale)
    for name, shp, _ in eng.variables():
        if name.endswith("moving_var"):
            eng.set_variable(name, np.ones(shp[0] * shp[1], np.float32))

    cpu_info = None
    if not dist_on and not args.no_cpu_baseline and args.mode == "infer":
        cpu_info = cpu_baseline(cfg)   # before the GPU legs: they then run back to back at the end of the process

    d = synth_pairs(B, npts, seed=1234 + rank, dtype=np.float32)
    p1 = torch.from_numpy(d["pcs1"]).to(dev)
    p2 = torch.from_numpy(d["pcs2"]).to(dev)
    nb2 = 2 * cfg["model"]["angles"]["num_bins"]
    outs = {k: torch.empty(B, nb2 if "logits" in k else 3, device=dev) for k in alignnet3d.OUTPUT_NAMES}
    ptrs = {k: v.data_ptr() for k, v in outs.items()}
    lab = {k: torch.from_numpy(np.ascontiguousarray(d[k])).to(dev) for k in
           ("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles")}
    lab_ptrs = {k: v.data_ptr() for k, v in lab.items()}
    want_train = args.mode == "train" or not args.no_train_leg
    rccl_ranks = 0

    def init_rccl():
        """Data-parallel training: RCCL communicator over xGMI inside the library; the 128-byte id travels via torch.distributed."""
        nonlocal rccl_ranks
        from alignnet3d import parallel
        parallel.init_comm(eng, dist, make_id=type(eng).comm_loopback_id if threads is not None else None, grad_communicator=bool(args.grad_communicator))
        eng.set_option("allreduce_overlap", args.allreduce_overlap)
        if args.sync_bn:
            eng.set_option("sync_bn", 1); eng.set_option("global_loss", 1)
        rccl_ranks = eng.get_option("comm_world")
        if rccl_ranks != world:
            raise RuntimeError(f"RCCL communicator reports {rccl_ranks} ranks, expected {world}")

    # which BatchNorm / loss semantics the training legs run in: "sync" = statistics and loss over the GLOBAL batch (the reference's
    # single-device step at world x batch; engine options sync_bn + global_loss), "local" = every rank a reference run on its own shard
    bn_mode = "sync" if (args.sync_bn and dist_on) else "local"

    def per_rank_rates(per_rank_seconds, steps):
        r = [B * steps / max(x, 1e-9) for x in per_rank_seconds]
        return {"min": round(min(r), 1), "max": round(max(r), 1), "ranks": len(r)}

    def sync_bn_fields():
        """sync-BN mode on > 1 rank: the step's dependency-bound per-layer all-reduces (30 + 17 gathers) each cost one small
        collective's latency; measured here with the same message size on the process group, the product is the floor they add to a step."""
        if bn_mode != "sync":
            return {}
        n = eng.get_option("sync_collectives")
        t = torch.zeros(2048, device=dev)
        for _ in range(20):
            dist.all_reduce(t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            dist.all_reduce(t)
        torch.cuda.synchronize()
        lat = (time.perf_counter() - t0) / 200
        return {"sync_collectives_per_step": n, "small_allreduce_latency_us": round(lat * 1e6, 2), "sync_bn_latency_floor_ms": round(n * lat * 1e3, 4)}

    if dist_on and args.mode == "train":
        init_rccl()   # the headline leg itself needs it: a failure here is fatal

    def train_step():
        eng.train_step_device(p1.data_ptr(), p2.data_ptr(), lab_ptrs, B)

    def infer_step():
        eng.forward_device(p1.data_ptr(), p2.data_ptr(), B, ptrs)

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ranks(x):
        """every rank's value of a scalar, in rank order (a straggler GPU shows up as a spread between min and max)"""
        if dist is None:
            return [x]
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def time_leg(step, steps, warmup, timers=True):
        """warmup untimed steps, then exactly `steps` steps between two fences; max over ranks.  Returns (seconds, kernel profile).
        timers: the engine's per-kernel HIP-event timers run inside the region (what `roofline` divides by).  They perturb what they time --
        a marker + a completion signal per timed kernel, ~7 us each: 0.02 ms on an inference step, 0.06 ms on a bf16 training step --
        so the secondary training legs time their `value` in a region of their own with the timers off."""
        for _ in range(warmup):
            step()
        fence()
        if timers:
            eng.profile_enable(True)
            eng.profile_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        kern = None
        if timers:
            kern = eng.profile_kernels()
            eng.profile_read(reset=True)
            eng.profile_enable(False)
        time_leg.last_per_rank = all_ranks(dt)   # this rank's own wall time of the region, from every rank
        return max_over_ranks(dt), kern

    def steps_for(step, floor):
        """Number of steps that fills --min-leg-seconds (secondary legs), from two timed probe steps."""
        step(); fence()
        t0 = time.perf_counter()
        step(); step(); fence()
        est = max_over_ranks((time.perf_counter() - t0) / 2)
        return max(floor, int(math.ceil(args.min_leg_seconds / max(est, 1e-6))))

    def roofline(kern, steps, bf16, leg_tag, backbone_name):
        """The leg's dominant timed kernel against the matrix-pipe roofline."""
        if not kern:
            return None
        cand = {k: v for k, v in kern.items() if k not in ("allreduce", "optimizer")}
        name = max(cand, key=lambda k: cand[k][0])
        ms, launches = cand[name]
        flops = 2.0 * kernel_macs(cfg, name, bf16) * 2 * B
        # the bf16 MFMA pipe is the bound of a kernel whose dominant products run on it: backbone (split), phase 2/3 and B2 in bf16 mode
        on_bf16 = bf16 and name in ("backbone", "train_fwd_phase2", "train_fwd_phase3", "train_bwd_b2")
        peak = (PEAK_BF16 / 3.0 if name == "backbone" else PEAK_BF16) if on_bf16 else PEAK_F32
        ms_step = ms / steps
        ach = flops / (ms_step * 1e-3) / 1e12 if ms_step > 0 and flops > 0 else None
        prof_name = backbone_name if name == "backbone" else KERNEL_IN_PROFILE.get(name, name)
        tr = pmc_traffic(leg_tag, prof_name, (B, npts))
        r = {"bound": "mfma", "achieved": None if ach is None else round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
             "frac": None if ach is None else round(ach / peak, 4),
             # HBM bytes of this kernel from the committed PMC passes ((2 FETCH_SIZE + WRITE_SIZE) KiB), per launch like `achieved`'s work
             "traffic": None if tr is None else tr["bytes_per_launch"], "traffic_per_step": None if tr is None else tr["bytes_per_step"],
             "kernel": prof_name.split("|")[0], "launches_per_step": launches / steps, "kernel_ms_per_step": round(ms_step, 4),
             "avg_launch_us": round(ms / max(launches, 1) * 1e3, 2), "algorithmic_flops_per_step": flops,
             "step_share": {k: round(v[0] / steps, 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][0])}}
        if tr is not None:
            r["traffic_source"] = tr["source"]      # committed rocprofv3 --pmc summary of this same command and shape ...
            r["traffic_commit"] = tr["commit"]      # ... taken at this commit of the tree (not measured inside this run)
        if on_bf16 and name == "backbone":
            r["note"] = "peak = dense bf16 MFMA peak / 3 (three bf16 MFMAs per algorithmic fp32 product)"
        if name == "knn" or (name == "train_bwd_b2" and bf16 and not dg):
            # VALU-issue-bound kernels (~1.1 k vector instructions per wave-tile against 40 MFMAs in pass B2's bf16
            # form, ~700 per kNN query): the matrix pipe is not what bounds them, so the roofline quoted is the vector-instruction issue
            # rate -- wave-level VALU instructions per launch / the launch's duration against 1228.8 G instructions / s.  Instruction
            # count: SQ_INSTS_VALU of the committed PMC pass of this command and shape, else the static per-tile count (`instr_source`).
            pv = pmc_valu_instr(leg_tag, prof_name, (B, npts))
            per_launch = pv["instr_per_launch"] if pv else VALU_INSTR_STATIC[name](npts) * 2 * B / max(launches / steps, 1)
            g = per_launch * (launches / steps) / (ms_step * 1e-3) / 1e9 if ms_step > 0 else None
            r["matrix_pipe"] = {"achieved": r["achieved"], "peak": r["peak"], "unit": "TFLOP/s", "frac": r["frac"]}   # kept for continuity with rounds 1-4
            r.update({"bound": "valu", "achieved": None if g is None else round(g, 1), "peak": PEAK_VALU_GINSTR, "unit": "Ginstr/s",
                      "frac": None if g is None else round(g / PEAK_VALU_GINSTR, 4), "valu_instr_per_launch": per_launch,
                      "instr_source": pv["source"] if pv else "static count (bench.py VALU_INSTR_STATIC, DESIGN.md 4.4 / 4.5)"})
        if name == "train_fwd_phase3":
            # extra passes never count as algorithmic work.  The Gram of h2 accumulated in this pass is this design's
            # substitute for the dense dW of the lift, so both fractions are quoted: with it (`frac`) and on the lift alone.
            lift = 2.0 * kernel_macs(cfg, name, bf16, with_gram=False) * 2 * B
            r["algorithmic_flops_per_step_lift_only"] = lift
            r["frac_lift_only"] = None if ach is None else round(lift / (ms_step * 1e-3) / 1e12 / peak, 4)
        return r

    # ---------------------------------------------------------------------------------------------------------- headline leg
    if args.mode == "infer" and args.infer_dtype == "bf16x3":
        eng.set_option("infer_matmul_bf16x3", 1)
    if args.mode == "train" and args.train_dtype == "bf16":
        eng.set_option("train_matmul_bf16", 1)
    executed = [0]   # every call of the headline step in this process: what a rocprofv3 run of this command divides its dispatch counts by

    def step():
        executed[0] += 1
        (train_step if args.mode == "train" else infer_step)()
    # clock spin-up, untimed and in addition to the W warm-up steps: the GPU sits in its low-power state (sclk ~100 MHz) while the CPU
    # baseline runs, and a short (W + K)-step run would be timed on the ramp
    spin_t0, spinup_steps = time.perf_counter(), 0
    while time.perf_counter() - spin_t0 < 0.4:
        for _ in range(8):
            step()
        eng.synchronize()
        spinup_steps += 8
    dt, kern = time_leg(step, args.steps, args.warmup)
    head_per_rank = time_leg.last_per_rank
    dt_off, _ = time_leg(step, args.steps, 0, timers=False)   # (reported next to `value`, never instead of it: the same K steps without the kernel timers)
    head_bf16 = (args.mode == "train" and args.train_dtype == "bf16") or (args.mode == "infer" and args.infer_dtype == "bf16x3")
    backbone_kernel = eng.last_backbone_kernel().split("<")[0] if args.mode == "infer" else "train"
    if args.mode == "infer":
        leg_tag = {("pointnet", "f32"): "", ("pointnet", "bf16x3"): "split", ("dgcnn", "f32"): "dgcnn", ("dgcnn", "bf16x3"): "dgcnn_split"}[(args.workload, args.infer_dtype)]
    else:
        leg_tag = "train_dgcnn" if dg else ("train_bf16" if args.train_dtype == "bf16" else "train")
    head_roof = roofline(kern, args.steps, head_bf16, leg_tag, backbone_kernel)

    # ---- from here on the headline (`value`, `roofline`) is measured.  On real ranks (one process per GPU) the legs below are the first thing that ever runs over
    #      RCCL at world > 1 on this code base's behalf: a watchdog makes sure a leg that hangs there costs the leg, not the line.
    single_rank = sustained = split_info = train_info = pcie_info = extra = None
    head_sync_bn = {}
    line_lock, line_written = threading.Lock(), [False]

    def write_line(note=None):
        """rank 0: the one JSON line, from whatever legs have completed (all of them unless the watchdog calls)"""
        ms_per_step = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        line = {
            "metric": "point-cloud pairs/sec at N=%d (inference, eval-mode forward)" % npts,
            "value": round(value, 1), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("SynthCars widths inference, batch=%d pairs/GPU, N=%d, fp32 (BASELINE.json configs[1])" % (B, npts))
                       if not dg else
                       ("SynthCars widths, DGCNN edge-conv branch (k=20), inference, batch=%d pairs/GPU, N=%d, fp32 (BASELINE.json configs[4] shape)" % (B, npts)),
                       "pairs_per_gpu": B, "num_points": npts, "parallelism": f"batch-split x{world} (no collective)", "devices_used": world},
            "roofline": head_roof, "spinup_steps_untimed": spinup_steps,
            # every call of the headline step in this process (spin-up, warm-up, the K timed steps, their repeat with the timers off, the
            # sustained loop): tools/summarize_prof.py divides a profiled run's dispatch counts by THIS number
            "steps_executed": executed[0],
            "without_kernel_timers": {"value": round(world * B * args.steps / dt_off, 1), "ms_per_step": round(dt_off / args.steps * 1e3, 4),
                                      "what": "the same K steps again with the per-kernel HIP-event timers off (`value` is the region they run in)"},
            "whole_path_tflops": round(FLOPS_PER_PAIR_TOTAL * B * args.steps / dt / 1e12 * 1.0, 2)
            if world == 1 and not dg and npts == N_POINTS and args.mode == "infer" else None,
        }
        if args.mode == "train":
            line["metric"] = "point-cloud pairs/sec at N=%d (training step)" % npts
            line["dtype"] = args.train_dtype
            line["config"]["workload"] = ("KITTITrackletsCars-style training step (SynthCars widths), batch=%d pairs/GPU, N=%d, %s "
                                          "(BASELINE.json configs[2])" % (B, npts, args.train_dtype))
            if dg:
                line["config"]["workload"] = ("training step, DGCNN edge-conv branch (k=20, SynthCars widths), batch=%d pairs/GPU, N=%d, %s "
                                              "(BASELINE.json configs[4] shape)" % (B, npts, "bf16 forward convs + h1 Q2, rest fp32" if args.train_dtype == "bf16" else "f32"))
            line["config"]["parallelism"] = f"data parallel x{world}" + (" (RCCL all-reduce of gradients, %d ranks)" % rccl_ranks if dist_on else "")
            line["bn_mode"] = bn_mode
            line["flops_per_pair_survey"] = FLOPS_PER_PAIR_TRAIN_SURVEY if not dg and npts == N_POINTS else None
            if dist_on:
                line["allreduce_exposed_ms_per_step"] = round(kern.get("allreduce", (0.0, 0))[0] / args.steps, 4)
                line.update(head_sync_bn)
        if dist_on:
            line["per_rank_pairs_per_s"] = per_rank_rates(head_per_rank, args.steps)
        if single_rank is not None:
            line["single_rank_reference"] = single_rank
            line["scaling_efficiency_vs_single_rank"] = round(value / (world * single_rank["value"]), 4)
        if threads is not None:
            line["n_gpus"] = 1
            line["config"]["devices_used"] = 1
            line["rehearsal"] = ("%d ranks as host threads on ONE GPU, in-process loopback communicator in place of RCCL, thread rendezvous in place of torch.distributed: "
                                 "exercises the world > 1 code path; NOT a throughput measurement (the ranks share the device)" % world)
        if args.force_dist:
            line["forced_dist"] = "world-1 rehearsal of the multi-GPU code path (torch.distributed.run --nproc-per-node=1, NCCL process group, RCCL communicator of one rank)"
        if args.mode == "infer" and args.infer_dtype == "bf16x3":
            line["dtype"] = "bf16x3"
            line["metric"] += " [split-bf16 backbone]"
            line["config"]["workload"] = line["config"]["workload"].replace("fp32", "split-bf16 products with fp32 accumulate")
        if split_info is not None:
            line["infer_bf16x3"] = split_info
        if train_info is not None:
            line["train"] = train_info
        if pcie_info is not None:
            line["pcie_inclusive"] = pcie_info
        if sustained is not None:
            line["sustained"] = sustained
        if cpu_info is not None:
            line["cpu_baseline"] = cpu_info
        if extra is not None:
            line.update(extra_seconds=extra.pop("seconds"), **extra)
        if note is None:
            line["options"] = engine_options(eng)   # as the legs left them (allreduce_overlap, sync_bn, ... are set after the engine is created)
        else:
            line["incomplete"] = note   # (written from the watchdog thread: the engine may be inside the call that hangs -- not asked for its options)
        with line_lock:
            if line_written[0]:
                return
            line_written[0] = True
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(line) + "\n").encode())

    watchdog = None
    if dist is not None and (world > 1 or args.force_dist) and threads is None and args.secondary_timeout > 0:
        def fire():
            if rank == 0:
                try:
                    write_line(note="watchdog: the legs after the headline measurement did not finish within %.0f s; the headline fields were measured normally, "
                                    "sections that had completed are included" % args.secondary_timeout)
                except Exception as e:   # noqa: BLE001
                    sys.stderr.write("bench.py watchdog: could not write the line: %s\n" % e)
            else:
                time.sleep(2.0)   # rank 0 writes first
            os._exit(0 if rank else (0 if line_written[0] else 1))
        watchdog = threading.Timer(args.secondary_timeout, fire)
        watchdog.daemon = True
        watchdog.start()
        if os.environ.get("BENCH_TEST_HANG_AFTER_HEADLINE"):   # test hook (tests/test_bench_contract_gpu.py): stand for a collective that never returns
            time.sleep(3600)

    # ---- world > 1: what ONE rank of this very run does with the node to itself -- the N = 1 value this line scales from (rank 0 steps alone, the
    #      other ranks wait at the fence), so that the scaling efficiency value / (n_gpus x this) is computable from this one line
    if dist is not None and world > 1:
        fence()
        if rank == 0:
            e1 = eng
            if args.mode == "train":   # (an engine of its own: the headline engine's step joins the communicator's collectives)
                e1 = alignnet3d.Engine(cfg, device=local_rank, seed=0)
                e1.set_option("train_matmul_bf16", int(args.train_dtype == "bf16"))
                step1 = lambda: e1.train_step_device(p1.data_ptr(), p2.data_ptr(), lab_ptrs, B)
            else:
                step1 = infer_step
            for _ in range(max(args.warmup, 3)):
                step1()
            e1.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step1()
            e1.synchronize()
            d1 = time.perf_counter() - t1
            single_rank = {"value": round(B * args.steps / d1, 1), "unit": "pairs/s", "ms_per_step": round(d1 / args.steps * 1e3, 4), "steps": args.steps, "n_gpus": 1,
                           "what": "rank 0 alone, the other ranks idle at a barrier: the same step" + (" on a fresh engine without a communicator (local-BN, no all-reduce)" if args.mode == "train" else "") +
                                   "; scaling efficiency of this line = value / (n_gpus x this value)"}
            if e1 is not eng:
                e1.close()
        fence()

    # ---- the same step, back to back for >= --sustained-seconds: the K-step region above is a burst of tens of milliseconds; this one
    #      is long enough for the clocks to settle (sclk sampled from rocm-smi while the queue is full, first and last chunk)
    if args.sustained_seconds > 0:
        per = max(dt / args.steps, 1e-5)
        chunk = max(1, int(0.25 / per))
        clk = [None, None]
        fence()
        n_s, t0 = 0, time.perf_counter()
        while True:
            for _ in range(chunk):
                step()
            n_s += chunk
            if clk[0] is None and rank == 0:
                clk[0] = sclk_mhz(local_rank) or 0
            eng.synchronize()
            last = time.perf_counter() - t0 >= args.sustained_seconds
            if dist is not None:   # every rank leaves the loop in the same round
                flag = torch.tensor([1.0 if last else 0.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                last = bool(flag.item() > 0)
            if last:
                for _ in range(chunk):
                    step()
                n_s += chunk
                if rank == 0:
                    clk[1] = sclk_mhz(local_rank)
                break
        fence()
        sdt_all = max_over_ranks(time.perf_counter() - t0)
        sustained = {"value": round(world * B * n_s / sdt_all, 1), "unit": "pairs/s", "ms_per_step": round(sdt_all / n_s * 1e3, 4), "steps": n_s,
                     "seconds": round(sdt_all, 2), "sclk_mhz_first_chunk": clk[0] or None, "sclk_mhz_last_chunk": clk[1],
                     "what": "the headline step back to back (chunks of %d steps, one host synchronisation and two rocm-smi reads in between); "
                             "`value` of the line stays the K-step figure" % chunk}

    # ------------------------------------------------- secondary leg: the opt-in split-bf16 backbone on the same batch (never the headline)
    if args.mode == "infer" and not dg and args.infer_dtype == "f32" and not args.no_split_leg:
        ref_out = {k: v.clone() for k, v in outs.items()}
        eng.set_option("infer_matmul_bf16x3", 1)
        ksteps = steps_for(infer_step, 10)
        sdt, skern = time_leg(infer_step, ksteps, 2)
        diff = max(float((outs[k] - ref_out[k]).abs().max().item()) for k in outs)
        split_info = {"value": round(world * B * ksteps / sdt, 1), "unit": "pairs/s", "ms_per_step": round(sdt / ksteps * 1e3, 4), "steps": ksteps,
                      "dtype": "bf16x3 (x = hi + lo bf16, three bf16 MFMAs per product, fp32 accumulate)",
                      "roofline": roofline(skern, ksteps, True, "split", eng.last_backbone_kernel().split("<")[0]),
                      "max_abs_diff_vs_exact_fp32_outputs": diff,
                      "what": "opt-in inference mode alignnet_set_option(infer_matmul_bf16x3); parity bar 1e-4 (tests/test_forward_gpu.py)"}
        eng.set_option("infer_matmul_bf16x3", 0)

    # ------------------------------- secondary leg: full training step (fwd with batch statistics + loss + bwd + all-reduce + Adam + EMA)
    if args.mode == "infer" and want_train:
      try:   # (a secondary leg must not cost the headline line: the communicator is created here, after the inference legs)
        if dist_on:
            init_rccl()
        train_info = {}
        for tdtype in ("f32", "bf16"):
            eng.set_option("train_matmul_bf16", int(tdtype == "bf16"))
            ksteps = steps_for(train_step, 5)
            tdt, _ = time_leg(train_step, ksteps, 1, timers=False)     # the leg's value: the step as a user runs it
            tdt_on, tkern = time_leg(train_step, ksteps, 1)              # the same K steps again under the kernel timers: roofline, step_share
            leg = {"value": round(world * B * ksteps / tdt, 1), "unit": "pairs/s", "ms_per_step": round(tdt / ksteps * 1e3, 3),
                   "ms_per_step_under_kernel_timers": round(tdt_on / ksteps * 1e3, 3), "steps": ksteps, "dtype": tdtype,
                   "roofline": roofline(tkern, ksteps, tdtype == "bf16", "train_bf16" if tdtype == "bf16" else "train", "train"),
                   "flops_per_pair_survey": FLOPS_PER_PAIR_TRAIN_SURVEY, "bn_mode": bn_mode,
                   "what": "train step: batch-stat forward + loss + backward + " +
                           ("RCCL all-reduce (%s) + " % ("3 buckets overlapped with the backward" if args.allreduce_overlap else "one call after the backward")
                            if dist_on else "") + "Adam + EMA, " + ("sync-BN + global-loss data parallel (the single-device step at the global batch)" if args.sync_bn else "local-BN data parallel") +
                           ("; MFMA convs on bf16 operands, fp32 accumulate (BASELINE.json configs[2])" if tdtype == "bf16" else "")}
            if dist_on:
                leg["rccl_ranks"] = rccl_ranks
                leg["allreduce_exposed_ms_per_step"] = round(tkern.get("allreduce", (0.0, 0))[0] / ksteps, 4)
                leg["per_rank_pairs_per_s"] = per_rank_rates(time_leg.last_per_rank, ksteps)
                leg.update(sync_bn_fields())
            if tdtype == "f32":
                train_info = leg
            else:
                train_info["bf16"] = leg
        eng.set_option("train_matmul_bf16", 0)
      except Exception as e:   # noqa: BLE001 -- reported in the line, the other legs stand
        train_info = {"error": "%s: %s" % (type(e).__name__, e)}

    # -------------- secondary leg: PCIe-inclusive inference, the reference's own timing methodology (train.py:447-449), rank 0's GPU only
    if args.mode == "infer" and not dg and args.infer_dtype == "f32" and not args.no_pcie_leg and rank == 0:
        for _ in range(2):
            eng.forward(d["pcs1"], d["pcs2"])
        kp = max(10, int(math.ceil(args.min_leg_seconds / 2.5e-3)))
        t0 = time.perf_counter()
        for _ in range(kp):
            eng.forward(d["pcs1"], d["pcs2"])
        pdt = time.perf_counter() - t0
        pcie_info = {"value": round(B * kp / pdt, 1), "unit": "pairs/s", "ms_per_step": round(pdt / kp * 1e3, 4), "steps": kp, "n_gpus": 1,
                     "what": "pageable host buffers in (2 x %.1f MB) and out, blocking alignnet_forward per batch -- the feed copy is inside the "
                             "timed region as in the reference's timing (train.py:447-449); not the headline value" % (B * npts * 12 / 1e6)}
        # the same host-to-host work through the pipelined path: pinned staging, copy-in of batch i + 1 under the forward of batch i
        for _ in range(3):
            eng.forward_submit(d["pcs1"], d["pcs2"]); eng.forward_wait()
        t0 = time.perf_counter()
        inflight = 0
        for _ in range(kp):
            if inflight == 2:
                eng.forward_wait(); inflight -= 1
            eng.forward_submit(d["pcs1"], d["pcs2"]); inflight += 1
        while inflight:
            eng.forward_wait(); inflight -= 1
        qdt = time.perf_counter() - t0
        pcie_info["pipelined"] = {"value": round(B * kp / qdt, 1), "unit": "pairs/s", "ms_per_step": round(qdt / kp * 1e3, 4), "steps": kp,
                                  "what": "alignnet_forward_submit / _wait: the same pageable buffers in and out, two batches in flight (pinned staging, "
                                          "H2D on a copy stream under the previous batch's forward, D2H on a third stream)"}
    # ---- short dgcnn / loader / icp legs (default one-GPU inference run only)
    if not dist_on and args.mode == "infer" and not dg and args.infer_dtype == "f32" and not args.no_extra_legs:
        extra = extra_legs(eng, local_rank, args.min_leg_seconds)
    # the headline training leg's sync-BN latency probe is a collective: EVERY rank runs it (only rank 0 writes the line)
    head_sync_bn = sync_bn_fields() if (dist_on and args.mode == "train") else {}
    if dist is not None:
        dist.barrier()

    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        write_line()
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
