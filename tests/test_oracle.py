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
