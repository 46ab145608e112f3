import sys, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/alignnet-3d_amd"]
import alignnet3d
from oracle import alignnet_ref as R, alignnet_torch as T
from tests.helpers import oracle_params, varied_pairs
B, N = 64, 512
cfg = alignnet3d.default_model_config(); cfg["model"]["num_points"] = N; cfg["training"]["batch_size"] = B
spec, P32 = oracle_params(cfg, seed=5)
d = varied_pairs(B, N, seed=5, dtype=np.float32)
rng = np.random.default_rng(5)
du = {k: rng.uniform(size=(B, 256)).astype(np.float32) for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")}
eng = alignnet3d.Engine(cfg); eng.set_variables(P32); eng.set_option("train_matmul_bf16", 1)
decay = eng.state()["bn_decay"]
eng.train_forward_backward(d["pcs1"], d["pcs2"], d, [du[k] for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")])
dec = eng.debug_train_decisions(B, relu=True); dec["round"] = eng.debug_train_rounded(B)
orig = T.TorchTp8._round_bf16_st
def hook(self, x, key=None):
    r = self.pinned["round"].get(key) if key else None
    if r is not None:
        xd = x.detach().numpy().ravel(); rr = np.asarray(r, np.float64).ravel()
        own = x.detach().to(torch.bfloat16).to(x.dtype).numpy().ravel()
        ulp = np.maximum(np.abs(xd), np.abs(rr)) * 2.0 ** -7 + 1e-5 * np.abs(xd).max()
        dd = np.abs(rr - xd) / ulp
        idx = np.argsort(-dd)[:4]
        print(key, "scale %.3g" % np.abs(xd).max(), "differ", int((rr != own).sum()), "worst:", [(int(i), float("%.6g" % xd[i]), float("%.6g" % rr[i]), float("%.3g" % dd[i])) for i in idx], flush=True)
    return orig(self, x, key)
T.TorchTp8._round_bf16_st = hook
tm = T.TorchTp8(spec, T.to_torch({k: v.astype(np.float64) for k, v in P32.items()}), bf16_lift=True, pinned=dec)
with torch.no_grad():
    tm.forward(torch.tensor(d["pcs1"].astype(np.float64)), torch.tensor(d["pcs2"].astype(np.float64)), True, decay, {k: torch.tensor(v.astype(np.float64)) for k, v in du.items()})
