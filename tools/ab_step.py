#!/usr/bin/env python3
"""Same-box A/B of the training step under engine options: alternates the option sets round-robin in ONE process (same clocks, same data) and
prints the median ms/step of each.    tools/ab_step.py [--bf16 1] [--dgcnn 1] [--batch 256] [--points 1024] [--rounds 5] [--steps 60] "k=v,k=v" "k=v" ...
An empty string "" is the default option set; "prof=1" switches the engine's HIP-event kernel timers on.  Options are alignnet_set_option keys (include/alignnet_hip.h)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "alignnet-3d_amd")]
import torch, alignnet3d
from alignnet3d.synth import synth_pairs
ap = argparse.ArgumentParser()
ap.add_argument("--bf16", type=int, default=1); ap.add_argument("--dgcnn", type=int, default=0); ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--points", type=int, default=1024); ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--infer", type=int, default=0)
ap.add_argument("sets", nargs="*", default=[""])
a = ap.parse_args()
cfg = alignnet3d.default_model_config(); cfg["model"]["num_points"] = a.points; cfg["training"]["batch_size"] = a.batch
if a.dgcnn: cfg["model"]["backbone"] = "dgcnn"
d = synth_pairs(a.batch, a.points, dtype=np.float32)
p1, p2 = torch.tensor(d["pcs1"]).cuda(), torch.tensor(d["pcs2"]).cuda()
lab = {k: torch.tensor(np.ascontiguousarray(d[k])).cuda() for k in ("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles")}
lp = {k: v.data_ptr() for k, v in lab.items()}
engs = []
for sset in a.sets:
    e = alignnet3d.Engine(cfg)
    for name, shp, _ in e.variables():
        if name.endswith("moving_var"): e.set_variable(name, np.ones(shp[0] * shp[1], np.float32))
    e.set_option("train_matmul_bf16", a.bf16)
    for kv in filter(None, sset.split(",")):
        k, v = kv.split("=")
        if k == "prof": e.profile_enable(bool(int(v)))     # (pseudo-option: the HIP-event kernel timers bench.py's timed region runs under)
        else: e.set_option(k, int(v))
    engs.append(e)
nb2 = 2 * cfg["model"]["angles"]["num_bins"]
outs = {k: torch.empty(a.batch, nb2 if "logits" in k else 3, device="cuda") for k in alignnet3d.OUTPUT_NAMES}; ptrs = {k: v.data_ptr() for k, v in outs.items()}
step = (lambda e: e.forward_device(p1.data_ptr(), p2.data_ptr(), a.batch, ptrs)) if a.infer else (lambda e: e.train_step_device(p1.data_ptr(), p2.data_ptr(), lp, a.batch))
for e in engs:
    for _ in range(20): step(e)
    e.synchronize()
res = [[] for _ in engs]
for r in range(a.rounds):
    for i, e in enumerate(engs):
        for _ in range(5): step(e)
        e.synchronize(); t = time.perf_counter()
        for _ in range(a.steps): step(e)
        e.synchronize(); res[i].append((time.perf_counter() - t) / a.steps * 1e3)
for sset, r in zip(a.sets, res):
    print("%-40s median %.4f ms/step  (min %.4f max %.4f)  %.1f k pairs/s" % (sset or "<default>", float(np.median(r)), min(r), max(r), a.batch / float(np.median(r))))
