# randomized agreement of the per-cloud training kernels dealt to several workgroups per cloud (options pn_cloud_parts / dg_cloud_parts: 0 = automatic, 2, 8) with one
# workgroup per cloud (= 1) over many (N, B): partial last tiles, fewer tiles than parts, B = 2, both backbones, fp32 and bf16 convs.  What differs is the grouping of partial
# sums (statistics, Grams, U2, Pdy); the folded running extremes of phase 3 are exact.  In bf16 mode stage 1's max-pool winners must be bit-equal (same points, fp64
# statistics partials, same rounded operands).  Bounds as tools/stress_tile_shapes.py.   usage: stress_cloud_parts.py [cases] [seed]
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'alignnet-3d_amd')]
import alignnet3d
from oracle import alignnet_ref as R
from tests.helpers import small_cfg, oracle_params
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
STD = dict(s1=(64, 128, 96), s2=(64, 128, 256), emb=(64, 128, 544))
bad = 0
for k in range(cases):
    N, B, bf16 = int(rng.integers(20, 900)), int(rng.integers(2, 40)), int(k % 2)
    backbone = "dgcnn" if k % 3 == 2 else "pointnet"
    if backbone == "dgcnn": N, B = max(N, 24), min(B, 12)
    cfg = small_cfg(N=N, nb=12, fc=(64, 32), backbone=backbone, **STD)
    cfg["training"]["batch_size"] = B
    spec, P32 = oracle_params(cfg, seed=k)
    d = R.synth_pairs(B, N, seed=k, dtype=np.float32)
    du = [rng.uniform(size=(B, 32)).astype(np.float32) for _ in range(5)]
    out = {}
    for parts in (1, 0, 2, 8):
        eng = alignnet3d.Engine(cfg); eng.set_variables(P32)
        eng.set_option("pn_cloud_parts", parts); eng.set_option("dg_cloud_parts", parts); eng.set_option("train_matmul_bf16", bf16)
        res = eng.train_forward_backward(d["pcs1"], d["pcs2"], d, du)
        out[parts] = (res, np.concatenate([eng.get_gradient(n).ravel() for n in R.trainable_names(spec)]), eng.debug_train_decisions(B)["pool"][0])
        eng.close()
    rb, gb, wb = out[1]
    for parts in (0, 2, 8):
        ra, ga, wa = out[parts]
        l2 = float(np.linalg.norm(ga - gb) / (np.linalg.norm(gb) + 1e-30))
        pe = max(float(np.abs(ra[k2] - rb[k2]).max()) for k2 in alignnet3d.OUTPUT_NAMES)
        moved = float((wa != wb).mean())
        # (free comparisons of two runs: one re-routed arg-max -- a max-pool winner or a dgcnn neighbour slot within rounding of a tie -- is 2e-2 of a dgcnn gradient at B ~ 10
        #  and up to half of a bf16 one at B <= 8; the sharp statement is tests/test_train_gpu.py's *_ragged cases: the split against the decision-pinned oracle, 1e-5 .. 5e-5)
        gl = (0.6 if B <= 8 else 0.3) if bf16 else (5e-2 if backbone == "dgcnn" else 2e-2)
        ok = np.isfinite(l2) and ((pe <= 5e-2 and l2 <= gl and (moved == 0.0 or backbone == "dgcnn")) if bf16 else (pe <= 2e-4 and l2 <= gl and moved <= 5e-3))
        bad += not ok
        print("%-8s %s N=%4d B=%2d parts %d: prediction diff %.2e, gradient rel L2 %.2e, stage-1 winners moved %.1e %s" % (backbone, "bf16" if bf16 else "fp32", N, B, parts, pe, l2, moved, "" if ok else "  <-- FAIL"), flush=True)
print("failures: %d of %d x 3" % (bad, cases))
sys.exit(1 if bad else 0)
