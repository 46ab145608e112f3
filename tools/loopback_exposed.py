#!/usr/bin/env python3
"""Stand-in for `allreduce_exposed_ms_per_step` on a 1-GPU box: W loopback ranks (one host thread + one engine each, DISTINCT shards) train on ONE device,
once with the bucketed all-reduce under the backward (allreduce_overlap = 1) and once with one all-reduce behind it (= 0), and report per rank and step
  * the wall time of a step (all W ranks share the GPU, so this is W shards' work, not a scaling number),
  * `allreduce`: the HIP-event time the compute stream waits at the join of the buckets (what bench.py prints as allreduce_exposed_ms_per_step), and
  * `comm_order` (362514 = every bucket issued right behind its stage's backward).
The loopback collective is a device kernel that moves the same bytes over HBM instead of xGMI: what carries over to RCCL is WHERE the wait sits, not its length.
    tools/loopback_exposed.py [--world 2] [--bf16 1] [--batch 256] [--points 1024] [--steps 30]"""
import argparse, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "alignnet-3d_amd")]
import torch, alignnet3d
from alignnet3d.synth import synth_pairs

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=2); ap.add_argument("--bf16", type=int, default=1); ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--points", type=int, default=1024); ap.add_argument("--steps", type=int, default=30); ap.add_argument("--sync-bn", type=int, default=0)
a = ap.parse_args()
W = a.world
cfg = alignnet3d.default_model_config(); cfg["model"]["num_points"] = a.points; cfg["training"]["batch_size"] = a.batch
LAB = ("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles")


def run(overlap):
    uid = alignnet3d.Engine.comm_loopback_id()
    out = [None] * W
    start = threading.Barrier(W)

    def worker(r):
        d = synth_pairs(a.batch, a.points, dtype=np.float32, seed=100 + r)       # every rank its own shard
        p1, p2 = torch.tensor(d["pcs1"]).cuda(), torch.tensor(d["pcs2"]).cuda()
        lab = {k: torch.tensor(np.ascontiguousarray(d[k])).cuda() for k in LAB}
        lp = {k: v.data_ptr() for k, v in lab.items()}
        e = alignnet3d.Engine(cfg)
        for name, shp, _ in e.variables():
            if name.endswith("moving_var"):
                e.set_variable(name, np.ones(shp[0] * shp[1], np.float32))
        e.set_option("train_matmul_bf16", a.bf16); e.set_option("allreduce_overlap", overlap); e.set_option("dropout_stream", r)
        if a.sync_bn:
            e.set_option("sync_bn", 1); e.set_option("global_loss", 1)
        e.comm_init(r, W, uid)
        for _ in range(5):
            e.train_step_device(p1.data_ptr(), p2.data_ptr(), lp, a.batch)
        e.synchronize(); start.wait()
        e.profile_enable(True); e.profile_read(reset=True)
        t = time.perf_counter()
        for _ in range(a.steps):
            e.train_step_device(p1.data_ptr(), p2.data_ptr(), lp, a.batch)
        e.synchronize(); dt = time.perf_counter() - t
        k = e.profile_kernels()
        out[r] = dict(ms=dt / a.steps * 1e3, allreduce=k.get("allreduce", (0.0, 0))[0] / a.steps, order=e.get_option("comm_order"), buckets=e.get_option("comm_buckets"))
        e.profile_enable(False); e.close()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    [t.start() for t in th]; [t.join() for t in th]
    return out


for overlap in (1, 0, 1, 0):
    res = run(overlap)
    print("allreduce_overlap=%d  world %d (one GPU): step %.3f ms  | compute stream waits %.3f ms/step at the join | comm_order %s, buckets %s" % (
        overlap, W, max(r["ms"] for r in res), float(np.mean([r["allreduce"] for r in res])), res[0]["order"], res[0]["buckets"]))
