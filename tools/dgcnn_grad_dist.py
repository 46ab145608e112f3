# DGCNN training: distribution of the per-tensor gradient error of the HIP step against the fp64 autograd oracle, next to the
# error of the SAME oracle evaluated in fp32 (how much of the difference is fp32 conditioning: near-tied max decisions etc.)
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'alignnet-3d_amd'))
import torch, alignnet3d
from oracle import alignnet_ref as R
from oracle import alignnet_torch as T
from tests.test_train_gpu import _setup_dgcnn, _setup, LABELS

def oracle(cfg, P32, d, du, decay, dt):
    spec = R.NetSpec.from_cfg(cfg)
    tp = T.to_torch({k: v.astype(dt) for k, v in P32.items()}, dtype=torch.float64 if dt == np.float64 else torch.float32, requires_grad=True)
    tm = T.TorchTp8(spec, tp)
    td = {k: torch.tensor(v.astype(dt)) for k, v in d.items()}
    tu = {k: torch.tensor(v.astype(dt)) for k, v in du.items()}
    ep = tm.forward(td["pcs1"], td["pcs2"], True, decay, tu)
    loss = tm.loss(ep, *[td[k] for k in LABELS]); loss.backward()
    return {k: (v.grad.numpy().astype(np.float64) if v.grad is not None else np.zeros(v.shape)) for k, v in tp.items() if v.requires_grad}

for (bk, N, B, seed, std) in [("dgcnn", 128, 16, 7, True), ("dgcnn", 128, 16, 8, True), ("dgcnn", 128, 32, 7, True), ("pointnet", 128, 16, 7, True)]:
    cfg, spec, P32, d, du = (_setup_dgcnn if bk == "dgcnn" else _setup)(N, B, seed=seed, std=std)
    eng = alignnet3d.Engine(cfg); eng.set_variables(P32)
    decay = eng.state()["bn_decay"]
    g64 = oracle(cfg, P32, d, du, decay, np.float64)
    g32 = oracle(cfg, P32, d, du, decay, np.float32)
    eng.train_forward_backward(d["pcs1"], d["pcs2"], d, [du[k] for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")])
    e_hip, e_t32 = [], []
    for name in R.trainable_names(spec):
        ref = g64[name]
        if np.abs(ref).max() < 1e-9: continue
        g = eng.get_gradient(name).astype(np.float64).reshape(ref.shape)
        e_hip.append(float(np.abs(g - ref).max() / np.abs(ref).max()))
        e_t32.append(float(np.abs(g32[name] - ref).max() / np.abs(ref).max()))
    e_hip, e_t32 = np.array(e_hip), np.array(e_t32)
    print("%s N %d B %d seed %d: HIP vs fp64 oracle  median %.1e p90 %.1e max %.1e   |   fp32 torch vs fp64 torch  median %.1e p90 %.1e max %.1e"
          % (bk, N, B, seed, np.median(e_hip), np.quantile(e_hip, .9), e_hip.max(), np.median(e_t32), np.quantile(e_t32, .9), e_t32.max()))
    eng.close()
