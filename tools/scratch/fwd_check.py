import sys, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/alignnet-3d_amd"]
import alignnet3d
from oracle import alignnet_ref as R, alignnet_torch as T
from tests.test_fullsize_gpu import _varied_setup
LABELS = ("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles")
cfg, spec, P32, d, du = _varied_setup()
eng = alignnet3d.Engine(cfg); eng.set_variables(P32)
decay = eng.state()["bn_decay"]
res = eng.train_forward_backward(d["pcs1"], d["pcs2"], d, [du[k] for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")])
tm = T.TorchTp8(spec, T.to_torch({k: v.astype(np.float64) for k, v in P32.items()}))
with torch.no_grad():
    td = {k: torch.tensor(v.astype(np.float64)) for k, v in d.items()}
    ep = tm.forward(td["pcs1"], td["pcs2"], True, decay, {k: torch.tensor(v.astype(np.float64)) for k, v in du.items()})
    loss = float(tm.loss(ep, *[td[k] for k in LABELS]))
print("loss engine %.9f oracle %.9f diff %.3e" % (res["loss"], loss, res["loss"] - loss))
for k in ep:
    e = res[k].astype(np.float64) - ep[k].numpy()
    print("  %-30s max |err| %.2e   mean err %+.2e (signed: a coherent shift shows here)   rms %.2e" % (k, np.abs(e).max(), e.mean(), np.sqrt((e ** 2).mean())))
