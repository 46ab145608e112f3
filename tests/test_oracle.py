"""CPU: the two independent restatements of the hot path must agree."""
import numpy as np
import pytest
import torch

from oracle import alignnet_ref as R
from oracle import alignnet_torch as T

LABELS = ("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles")


def _setup(backbone, N, B=4, seed=0):
    spec = R.NetSpec(num_points=N, backbone=backbone, num_bins=12, s1_conv=(16, 32, 48), s2_conv=(16, 32, 64),
                     emb_conv=(16, 32, 80), s1_fc=(32, 16), s2_fc=(32, 16), rem_fc=(32, 16), knn_k=5)
    P = R.init_params(spec, seed)
    R.randomize_bn(P)
    d = R.synth_pairs(B, N, dtype=np.float64)
    rng = np.random.default_rng(0)
    du = {k: rng.uniform(size=(B, 16)) for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")}
    return spec, P, d, du


@pytest.mark.parametrize("backbone,N", [("pointnet", 64), ("dgcnn", 24)])
@pytest.mark.parametrize("training", [False, True])
def test_numpy_vs_torch_fp64(backbone, N, training):
    spec, P, d, du = _setup(backbone, N)
    ep, upd, _ = R.get_model(P, spec, d["pcs1"], d["pcs2"], training, 0.5, du if training else None)
    loss, _ = R.get_loss(spec, ep, *[d[k] for k in LABELS])
    tm = T.TorchTp8(spec, T.to_torch(P))
    td = {k: torch.tensor(v) for k, v in d.items()}
    tu = {k: torch.tensor(v) for k, v in du.items()}
    tep = tm.forward(td["pcs1"], td["pcs2"], training, 0.5, tu if training else None)
    tl = tm.loss(tep, *[td[k] for k in LABELS])
    for k in ep:
        np.testing.assert_allclose(ep[k], tep[k].numpy(), rtol=1e-9, atol=1e-9)
    assert abs(loss - float(tl)) <= 1e-9 * max(1.0, abs(loss))
    if training:
        for k in upd:
            np.testing.assert_allclose(upd[k], tm.ema_updates[k].numpy(), rtol=1e-9, atol=1e-11)


def test_fp32_close_to_fp64():
    spec, P, d, _ = _setup("pointnet", 64)
    ep64, _, _ = R.get_model(P, spec, d["pcs1"], d["pcs2"])
    ep32, _, _ = R.get_model(R.cast_params(P, np.float32), spec, d["pcs1"].astype(np.float32), d["pcs2"].astype(np.float32))
    for k in ep64:
        assert ep32[k].dtype == np.float32
        np.testing.assert_allclose(ep32[k], ep64[k], rtol=1e-4, atol=1e-4)


def test_trainable_count_matches_survey():
    # SynthCars widths -> 2,165,073 trainable floats
    spec = R.NetSpec()
    shapes = dict(R.param_names(spec))
    assert sum(int(np.prod(shapes[n])) for n in R.trainable_names(spec)) == 2165073


def test_permutation_invariance_and_translation_equivariance():
    spec, P, d, _ = _setup("pointnet", 64)
    ep, _, _ = R.get_model(P, spec, d["pcs1"], d["pcs2"])
    perm = np.random.default_rng(3).permutation(64)
    ep_p, _, _ = R.get_model(P, spec, d["pcs1"][:, perm], d["pcs2"][:, perm])
    for k in ep:
        np.testing.assert_allclose(ep[k], ep_p[k], rtol=1e-9, atol=1e-9)
    shift = np.array([3.0, -2.0, 0.5])
    ep_s, _, _ = R.get_model(P, spec, d["pcs1"] + shift, d["pcs2"] + shift)
    for k in ("pred_s1_pc1centers", "pred_s2_pc1centers", "pred_s1_pc2centers", "pred_s2_pc2centers"):
        np.testing.assert_allclose(ep_s[k], ep[k] + shift, rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(ep_s["pred_translations"], ep["pred_translations"], rtol=1e-9, atol=1e-8)


def test_autograd_matches_finite_differences():
    spec, P, d, du = _setup("pointnet", 32, B=3)
    tp = T.to_torch(P, requires_grad=True)
    tm = T.TorchTp8(spec, tp)
    td = {k: torch.tensor(v) for k, v in d.items()}
    tu = {k: torch.tensor(v) for k, v in du.items()}

    def f():
        return tm.loss(tm.forward(td["pcs1"], td["pcs2"], True, 0.5, tu), *[td[k] for k in LABELS])

    f().backward()
    rng = np.random.default_rng(0)
    for name in ("siamese/transformer1/embedding/conv2/weights", "siamese_1/embedding/conv3/bn/gamma", "fc3/biases",
                 "siamese/transformer2/mlp/fc1/weights"):
        p = tp[name]
        idx = tuple(int(rng.integers(0, s)) for s in p.shape)
        eps = 1e-6
        with torch.no_grad():
            old = p[idx].item()
            p[idx] = old + eps
            lp = float(f())
            p[idx] = old - eps
            lm = float(f())
            p[idx] = old
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - p.grad[idx].item()) <= 1e-5 * max(1.0, abs(fd)), (name, fd, p.grad[idx].item())


def test_schedules_and_adam():
    # train.py:133-174 with SynthCars-like numbers
    assert R.learning_rate(0, 128, 20000, 0.005, 30, 0.5) == 0.005
    steps_per_epoch = 20000 // 128
    assert R.learning_rate(30 * steps_per_epoch, 128, 20000, 0.005, 30, 0.5) == 0.0025
    assert R.learning_rate(10 ** 9, 128, 20000, 0.005, 30, 0.5) == 1e-5
    assert R.bn_decay_schedule(0, 128, 20000, 0.5, 30, 0.5, 0.99) == 0.5
    assert R.bn_decay_schedule(30 * steps_per_epoch, 128, 20000, 0.5, 30, 0.5, 0.99) == 0.75
    assert R.bn_decay_schedule(10 ** 9, 128, 20000, 0.5, 30, 0.5, 0.99) == 0.99
    w, m, v = R.adam_step(np.array([1.0]), np.array([0.5]), np.zeros(1), np.zeros(1), 1, 0.01)
    # first TF-Adam step moves by ~lr*sign(g)
    assert abs((1.0 - w[0]) - 0.01) < 1e-6


def _loss_and_grads(tm, tp, td, tu):
    for v in tp.values():
        v.grad = None
    loss = tm.loss(tm.forward(td["pcs1"], td["pcs2"], True, 0.5, tu), *[td[k] for k in LABELS])
    loss.backward()
    return float(loss.detach()), {k: v.grad.numpy().copy() for k, v in tp.items() if v.grad is not None}


@pytest.mark.parametrize("backbone,N", [("pointnet", 64), ("dgcnn", 24)])
def test_pinned_to_own_decisions_changes_nothing(backbone, N):
    """Decision-pinned mode (oracle/alignnet_torch.py `pinned`): gathering at the oracle's OWN max-pool winners, neighbour slots,
    neighbour table and yaw classes is the same function -- loss and every gradient equal the unpinned evaluation's, every
    reported gap is zero.  (The tower layout of `pinned` is the engine's: [tower][...].)"""
    spec, P, d, du = _setup(backbone, N)
    td = {k: torch.tensor(v) for k, v in d.items()}
    tu = {k: torch.tensor(v) for k, v in du.items()}
    tp = T.to_torch(P, requires_grad=True)
    tm = T.TorchTp8(spec, tp)
    tm.record_decisions = True
    loss0, g0 = _loss_and_grads(tm, tp, td, tu)
    dec = tm.decisions
    pinned = {"yaw": np.stack(dec["yaw"]), "pool": [np.stack(x) for x in dec["pool"]]}
    if backbone == "dgcnn":
        pinned["slot"] = [np.stack(x) for x in dec["slot"]]
        pinned["knn"] = np.stack(dec["knn"])
    tm2 = T.TorchTp8(spec, tp, pinned=pinned)
    loss1, g1 = _loss_and_grads(tm2, tp, td, tu)
    assert loss1 == loss0
    for k in g0:
        np.testing.assert_allclose(g1[k], g0[k], rtol=1e-12, atol=1e-15, err_msg=k)
    assert tm2.pin_report and all(gap == 0.0 and differ == 0 for _, gap, _, differ, _ in tm2.pin_report), tm2.pin_report
    kinds = {r[0].split(":")[0] for r in tm2.pin_report}
    assert kinds == ({"yaw", "pool"} if backbone == "pointnet" else {"yaw", "pool", "slot", "knn"})


def test_pinned_evaluation_is_continuous_where_the_free_one_jumps():
    """Why the pin exists: move the inputs by 1e-7 and the free evaluation re-decides near-tied winners (its gradient moves by per cents
    at full size); pinned to the unperturbed decisions the same perturbation moves the gradient by ~1e-6, and the report shows the
    pinned winners are maxima of the perturbed values to within that perturbation.  A WRONG pin (a winner that is not a maximum) shows
    up as a gap far above rounding."""
    spec, P, d, du = _setup("pointnet", 64, B=6)
    td = {k: torch.tensor(v) for k, v in d.items()}
    tu = {k: torch.tensor(v) for k, v in du.items()}
    tp = T.to_torch(P, requires_grad=True)
    tm = T.TorchTp8(spec, tp)
    tm.record_decisions = True
    _, g0 = _loss_and_grads(tm, tp, td, tu)
    pinned = {"yaw": np.stack(tm.decisions["yaw"]), "pool": [np.stack(x) for x in tm.decisions["pool"]]}
    rng = np.random.default_rng(1)
    td2 = dict(td)
    for k in ("pcs1", "pcs2"):
        td2[k] = td[k] + torch.tensor(rng.normal(scale=1e-7, size=td[k].shape))
    tm2 = T.TorchTp8(spec, tp, pinned=pinned)
    _, g1 = _loss_and_grads(tm2, tp, td2, tu)
    gs = max(np.abs(v).max() for v in g0.values())
    worst = max(np.abs(g1[k] - g0[k]).max() / gs for k in g0)
    assert worst < 1e-4, worst
    assert all(gap <= 1e-5 * max(scale, 1.0) for _, gap, scale, _, _ in tm2.pin_report), tm2.pin_report
    bad = {"yaw": pinned["yaw"], "pool": [x.copy() for x in pinned["pool"]]}
    bad["pool"][2][0, 0, :] = (bad["pool"][2][0, 0, :] + 17) % 64   # tower 0, pair 0 of the embedding stage: every winner moved to another point
    tm3 = T.TorchTp8(spec, tp, pinned=bad)
    _loss_and_grads(tm3, tp, td, tu)
    gap = max(g for what, g, _, _, _ in tm3.pin_report if what == "pool:embedding:0")
    assert gap > 1e-3, gap
