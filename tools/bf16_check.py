# bf16-lift training step vs the fp32 step on the same inputs: decode flips, loss, gradient cosines (GPU)
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'alignnet-3d_amd')); sys.path.insert(0, ROOT)
import alignnet3d
from oracle import alignnet_ref as R
from tests.test_train_gpu import _setup
nb = 12
for (N, B) in [(256, 16), (256, 64), (256, 128)]:
    for seed in (5, 6):
        cfg, spec, P32, d, du = _setup(N, B, seed=seed)
        us = None
        out = {}
        for mode in (0, 1):
            eng = alignnet3d.Engine(cfg); eng.set_variables(P32); eng.set_option("train_matmul_bf16", mode)
            rng = np.random.default_rng(seed)
            us = [rng.uniform(size=(B, 32)).astype(np.float32) for _ in range(5)]
            r = eng.train_forward_backward(d["pcs1"], d["pcs2"], d, us)
            out[mode] = (r, {n: eng.get_gradient(n).astype(np.float64).ravel() for n in R.trainable_names(spec)}); eng.close()
        (r32, g32), (r16, g16) = out[0], out[1]
        flips = sum(int((np.argmax(r32[k][:, :nb], 1) != np.argmax(r16[k][:, :nb], 1)).sum()) for k in ("pred_pc1angle_logits", "pred_pc2angle_logits"))
        a = np.concatenate([g32[n] for n in g32]); b = np.concatenate([g16[n] for n in g32])
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
        cs = {n: float(g32[n] @ g16[n] / (np.linalg.norm(g32[n]) * np.linalg.norm(g16[n]) + 1e-300)) for n in g32 if np.abs(g32[n]).max() > 1e-3 * np.abs(a).max() and g32[n].size >= 32}
        w = sorted(cs.items(), key=lambda kv: kv[1])[:3]
        print(f"N={N} B={B} seed={seed} flips={flips} loss {r32['loss']:.5f} {r16['loss']:.5f} cos_all={cos:.4f} worst={[(k[-40:], round(v, 3)) for k, v in w]}", flush=True)
