#!/usr/bin/env python3
"""Headline benchmark: AlignNet-3D point-cloud pairs/sec at N=1024 on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode infer|train] [--batch B]

A "step" is one pass of the hot path (models/tp8.py get_model, eval mode, i.e. the
reference's timed `sess.run`, train.py:447-449) over one batch of synthetic pairs that is
already resident in HBM.  Workload at N=1 = BASELINE.json configs[1]: SynthCars widths,
batch 256, N=1024, fp32.  With --gpus N (launched under torch.distributed.run, one rank per
GPU) every rank processes its own batch of 256 pairs (weak scaling; pairs are independent in
eval mode so there is no data-path collective); value = all pairs / max-over-ranks time.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     dominant kernel = pointnet_fused (the fused shared-MLP backbone).  achieved =
               algorithmic FLOPs per step of that kernel (2 * 260,636,672 MAC per
               cloud-triple * 2 clouds per pair) / its HIP-event time on the engine's stream,
               measured inside the timed region.  peak = 157.3 TFLOP/s fp32 MFMA
               (MI355X_MICROARCH.md).  traffic = HBM bytes per step from rocprofv3 PMC
               (profiles/*_pmc_traffic.json: the dominant kernel's three launches of one step, collected
               in separate --pmc passes), else null.
  cpu_baseline the oracle ("port": unfused op-by-op NumPy fp32 restatement, eval mode) timed on
               this box's host cores on a bounded sample (batch 32, repeated ~10-20 s).
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "alignnet-3d_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

# algorithmic work (SURVEY.md 8d / BASELINE.md 2), SynthCars widths, N=1024, nb=50
FLOPS_PER_PAIR_TOTAL = 1_047_688_704
N_POINTS = 1024


def backbone_macs_per_cloud(cfg):
    """MACs of the three fused backbones for one cloud (the dominant kernel's algorithmic work).
    dgcnn (models/tp8.py:30-46): edge convs widths[:-1] on k = 20 edges per point, then widths[-1] per point."""
    o = cfg["model"]["options"]
    n = cfg["model"]["num_points"]
    dg = cfg["model"]["backbone"] == "dgcnn"
    tot = 0
    for widths in (o["s1transformer"][0], o["s2transformer"][0], o["embedding"]):
        if dg:
            cin, edge = 6, 0
            for c in widths[:-1]:
                edge += cin * c
                cin = c
            tot += 20 * edge + cin * widths[-1]
        else:
            cin = 3
            for c in widths:
                tot += cin * c
                cin = c
    return tot * n


def cpu_baseline(cfg, seconds=12.0):
    from oracle import alignnet_ref as R
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    spec = R.NetSpec.from_cfg(cfg)
    P = R.init_params(spec, 0, np.float32)
    for k in P:  # plausible eval-mode shadows: mean 0, var 1
        if k.endswith("moving_var"):
            P[k] = np.ones_like(P[k])
    B = 32  # the reference's own timing mode uses bs=32 (train.py:557)
    d = R.synth_pairs(B, spec.num_points, seed=1234, dtype=np.float32)
    R.get_model(P, spec, d["pcs1"], d["pcs2"])  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        R.get_model(P, spec, d["pcs1"], d["pcs2"])
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 200:
            break
    return {"value": round(B * n / dt, 2), "unit": "pairs/s", "cores": int(threads), "kind": "port",
            "sample": f"{n} x batch {B} (N={spec.num_points}, SynthCars widths, eval mode), unfused NumPy fp32 oracle, "
                      f"{dt:.1f} s wall, forward only (cf. reference train.py:447-449)"}


def pmc_traffic():
    """HBM bytes per step from a committed rocprofv3 PMC summary, if one exists."""
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))
                   if re.fullmatch(r"r\d+_pmc_traffic\.json", os.path.basename(f)))   # the exact-fp32 inference profile of a round
    if not files:
        return None
    try:
        j = json.load(open(files[-1]))
        return j.get("backbone_hbm_bytes_per_step") or j.get("hbm_bytes_per_step")
    except Exception:
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
    ap.add_argument("--train-leg", action="store_true", help="also run the training leg when --gpus > 1 (RCCL all-reduce)")
    ap.add_argument("--workload", choices=["pointnet", "dgcnn"], default="pointnet",
                    help="dgcnn = BASELINE.json configs[4] shape: N=4096, edge-conv branch (--mode train: fp32 only)")
    ap.add_argument("--points", type=int, default=0, help="points per cloud (default 1024; 4096 for dgcnn)")
    args = ap.parse_args()

    import torch
    import alignnet3d
    from alignnet3d.synth import synth_pairs  # synthetic input generator (SURVEY 8d recipe); the oracle is only imported by cpu_baseline()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank %= max(torch.cuda.device_count(), 1)   # more ranks than GPUs: share devices rather than fail
        torch.cuda.set_device(local_rank)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        except Exception as e:   # the inference path has no data-path collective: a barrier and one max are all it needs
            print(f"[bench] nccl backend unavailable ({e}); using gloo for the barrier / max-over-ranks", file=sys.stderr, flush=True)
            dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cfg = alignnet3d.default_model_config()
    npts = args.points or (4096 if args.workload == "dgcnn" else N_POINTS)
    cfg["model"]["num_points"] = npts
    cfg["model"]["backbone"] = args.workload
    if args.workload == "dgcnn":
        args.no_train_leg = True
        args.no_cpu_baseline = True
    cfg["training"]["batch_size"] = args.batch * world
    B = args.batch
    eng = alignnet3d.Engine(cfg, device=local_rank, seed=0)
    # eval-mode shadows: mean 0 / var 1 (a freshly initialised net has var 0 -> degenerate scale)
    for name, shp, _ in eng.variables():
        if name.endswith("moving_var"):
            eng.set_variable(name, np.ones(shp[0] * shp[1], np.float32))

    d = synth_pairs(B, npts, seed=1234 + rank, dtype=np.float32)
    p1 = torch.from_numpy(d["pcs1"]).to(dev)
    p2 = torch.from_numpy(d["pcs2"]).to(dev)
    nb2 = 2 * cfg["model"]["angles"]["num_bins"]
    outs = {k: torch.empty(B, nb2 if "logits" in k else 3, device=dev) for k in alignnet3d.OUTPUT_NAMES}
    ptrs = {k: v.data_ptr() for k, v in outs.items()}

    lab = {k: torch.from_numpy(np.ascontiguousarray(d[k])).to(dev) for k in
           ("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles")}
    lab_ptrs = {k: v.data_ptr() for k, v in lab.items()}
    want_train = args.mode == "train" or (not args.no_train_leg and (world == 1 or args.train_leg))
    if world > 1 and want_train:
        # data-parallel training: RCCL communicator over xGMI inside the library; the 128-byte id travels via torch.distributed
        from alignnet3d import parallel
        parallel.init_comm(eng, dist)

    def train_step():
        eng.train_step_device(p1.data_ptr(), p2.data_ptr(), lab_ptrs, B)

    def infer_step():
        eng.forward_device(p1.data_ptr(), p2.data_ptr(), B, ptrs)

    step = train_step if args.mode == "train" else infer_step
    if args.mode == "infer" and args.infer_dtype == "bf16x3":
        eng.set_option("infer_matmul_bf16x3", 1)
    if args.mode == "train" and args.train_dtype == "bf16":
        eng.set_option("train_matmul_bf16", 1)

    def max_over_ranks(x):
        if dist is None:
            return x
        on_gpu = dist.get_backend() == "nccl"
        t = torch.tensor([x], device=dev if on_gpu else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.profile_enable(True)
    eng.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = eng.profile_read(reset=True)
    eng.profile_enable(False)

    dt = max_over_ranks(dt)

    # secondary leg: the opt-in split-bf16 backbone on the same batch (never the headline value)
    split_info = None
    if args.mode == "infer" and args.workload == "pointnet" and args.infer_dtype == "f32":
        ref_out = {k: v.clone() for k, v in outs.items()}
        eng.set_option("infer_matmul_bf16x3", 1)
        ksteps = max(10, args.steps // 2)
        for _ in range(3):
            infer_step()
        fence()
        eng.profile_enable(True)
        eng.profile_read(reset=True)
        t1 = time.perf_counter()
        for _ in range(ksteps):
            infer_step()
        fence()
        sdt = time.perf_counter() - t1
        sprof = eng.profile_read(reset=True)
        eng.profile_enable(False)
        sdt = max_over_ranks(sdt)
        diff = max(float((outs[k] - ref_out[k]).abs().max().item()) for k in outs)
        bb_ms = sprof["backbone_ms"] / ksteps
        split_info = {"value": round(world * B * ksteps / sdt, 1), "unit": "pairs/s", "ms_per_step": round(sdt / ksteps * 1e3, 4), "steps": ksteps,
                      "dtype": "bf16x3 (x = hi + lo bf16, three bf16 MFMAs per product, fp32 accumulate)",
                      "kernel_ms_per_step": round(bb_ms, 4),
                      "bf16_mfma_tflops": round(3 * 2.0 * backbone_macs_per_cloud(cfg) * 2 * B / (bb_ms * 1e-3) / 1e12, 1) if bb_ms > 0 else None,
                      "bf16_mfma_peak_tflops": 2500.0,
                      "max_abs_diff_vs_exact_fp32_outputs": diff,
                      "what": "opt-in inference mode alignnet_set_option(infer_matmul_bf16x3); parity bar 1e-4 (tests/test_forward_gpu.py)"}
        eng.set_option("infer_matmul_bf16x3", 0)

    # secondary leg: full training step (fwd with batch statistics + loss + bwd + all-reduce + Adam + EMA), fp32
    train_info = None
    if args.mode == "infer" and want_train:
        train_info = {}
        for tdtype in ("f32", "bf16"):
            eng.set_option("train_matmul_bf16", int(tdtype == "bf16"))
            ksteps = max(3, args.steps // 5)
            for _ in range(2):
                train_step()
            fence()
            t1 = time.perf_counter()
            for _ in range(ksteps):
                train_step()
            fence()
            tdt = time.perf_counter() - t1
            tdt = max_over_ranks(tdt)
            leg = {"value": round(world * B * ksteps / tdt, 1), "unit": "pairs/s", "ms_per_step": round(tdt / ksteps * 1e3, 3),
                   "steps": ksteps, "dtype": tdtype,
                   "what": "train step: batch-stat forward + loss + backward + " +
                           ("RCCL all-reduce + " if world > 1 else "") + "Adam + EMA, local-BN data parallel" +
                           ("; 128->C3 lift and its Gram on bf16 MFMA, fp32 accumulate (BASELINE.json configs[2])" if tdtype == "bf16" else "")}
            if tdtype == "f32":
                train_info = leg
            else:
                train_info["bf16"] = leg
        eng.set_option("train_matmul_bf16", 0)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        bb_flops_per_step = 2.0 * backbone_macs_per_cloud(cfg) * 2 * B
        bb_ms_per_step = prof["backbone_ms"] / args.steps
        achieved = bb_flops_per_step / (bb_ms_per_step * 1e-3) / 1e12 if bb_ms_per_step > 0 else None
        peak = 157.3
        line = {
            "metric": "point-cloud pairs/sec at N=%d (inference, eval-mode forward)" % npts,
            "value": round(value, 1), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("SynthCars widths inference, batch=%d pairs/GPU, N=%d, fp32 (BASELINE.json configs[1])" % (B, npts))
                       if args.workload == "pointnet" else
                       ("SynthCars widths, DGCNN edge-conv branch (k=20), inference, batch=%d pairs/GPU, N=%d, fp32 (BASELINE.json configs[4] shape)" % (B, npts)),
                       "pairs_per_gpu": B, "num_points": npts, "parallelism": f"batch-split x{world} (no collective)"},
            "roofline": {"bound": "mfma", "achieved": None if achieved is None else round(achieved, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": None if achieved is None else round(achieved / peak, 4),
                         "traffic": pmc_traffic() if args.workload == "pointnet" and npts == N_POINTS and B == 256 else None,
                         "kernel": "pointnet_fused" if args.workload == "pointnet" else "dgcnn_fused",
                         "launches_per_step": prof["backbone_launches"] / args.steps,
                         "kernel_ms_per_step": round(bb_ms_per_step, 4),
                         "algorithmic_flops_per_step": bb_flops_per_step},
            "whole_path_tflops": round(FLOPS_PER_PAIR_TOTAL * B * args.steps / dt / 1e12 * 1.0, 2)
            if world == 1 and args.workload == "pointnet" and npts == N_POINTS else None,
        }
        if args.mode == "train":
            line["metric"] = "point-cloud pairs/sec at N=%d (training step)" % npts
            line["roofline"] = None
            line["dtype"] = args.train_dtype
            line["config"]["workload"] = ("KITTITrackletsCars-style training step (SynthCars widths), batch=%d pairs/GPU, N=%d, %s "
                                          "(BASELINE.json configs[2])" % (B, npts, args.train_dtype))
            if args.workload == "dgcnn":
                line["config"]["workload"] = ("training step, DGCNN edge-conv branch (k=20, SynthCars widths), batch=%d pairs/GPU, N=%d, f32 "
                                              "(BASELINE.json configs[4] shape)" % (B, npts))
            line["config"]["parallelism"] = f"data parallel x{world}" + (" (RCCL all-reduce of gradients)" if world > 1 else "")
            line["whole_path_tflops"] = None
        if args.mode == "infer" and args.infer_dtype == "bf16x3":
            line["dtype"] = "bf16x3"
            line["metric"] += " [split-bf16 backbone]"
            line["config"]["workload"] = line["config"]["workload"].replace("fp32", "split-bf16 products with fp32 accumulate")
            line["roofline"]["peak"] = 2500.0 / 3.0
            line["roofline"]["frac"] = None if achieved is None else round(achieved / (2500.0 / 3.0), 4)
            line["roofline"]["kernel"] = "pointnet_split" if args.workload == "pointnet" else "dgcnn_split"
            line["roofline"]["traffic"] = None
            line["roofline"]["note"] = "peak = dense bf16 MFMA peak / 3 (three bf16 MFMAs per algorithmic fp32 product)"
        if split_info is not None:
            line["infer_bf16x3"] = split_info
        if train_info is not None:
            line["train"] = train_info
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
