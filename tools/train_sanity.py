# training sanity on synthetic pairs: mean loss of the first / last 10 of K steps, fp32 and bf16 convs, PointNet and dgcnn, shipped widths
# (the kernels of the default dispatch: 128-point phase 3, fused Gram, dense edge backward ...); usage: train_sanity.py [K]
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'alignnet-3d_amd')]
import alignnet3d
from oracle import alignnet_ref as R
K = int(sys.argv[1]) if len(sys.argv) > 1 else 120
for backbone in ("pointnet", "dgcnn"):
    for bf16 in (0, 1):
        cfg = alignnet3d.default_model_config()
        cfg["model"]["num_points"] = 320            # partial last tile in both tile shapes
        cfg["model"]["backbone"] = backbone
        cfg["training"]["batch_size"] = 32
        eng = alignnet3d.Engine(cfg, seed=1)
        eng.set_option("train_matmul_bf16", bf16)
        losses = []
        for k in range(K):
            d = R.synth_pairs(32, 320, seed=100 + k % 8, dtype=np.float32)   # eight batches, revisited: the loss must fall
            losses.append(eng.train_step(d["pcs1"], d["pcs2"], d)["loss"])
        print("%-8s %s: loss first 10 %.4f -> last 10 %.4f, finite %s, kernel mask %d" % (backbone, "bf16" if bf16 else "fp32", np.mean(losses[:10]), np.mean(losses[-10:]),
              bool(np.all(np.isfinite(losses))), eng.get_option("last_train_kernel")))
        eng.close()
