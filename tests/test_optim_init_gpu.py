"""GPU: SURVEY 8 rows a12 (optimiser + schedules as the ENGINE computes them) and a14 (variable initialisation), plus the
distribution of the device-side dropout stream (a5).  References: train.py:133-174 (schedules), :211-217 (optimisers),
utils/tf_util.py:10-49 (Xavier init), :470-480 (BN variables / EMA slots), :554-575 (dropout)."""
import math

import numpy as np
import pytest

import alignnet3d
from oracle import alignnet_ref as R
from tests.helpers import small_cfg, oracle_params

pytestmark = pytest.mark.gpu
US = ("s1_0", "s2_0", "s1_1", "s2_1", "rem")


def _batch(cfg, B, N, seed):
    d = R.synth_pairs(B, N, seed=seed, dtype=np.float32)
    rng = np.random.default_rng(seed)
    u = [rng.uniform(size=(B, 32)).astype(np.float32) for _ in US]
    return d, u


def _all(eng, names, getter):
    return {n: getter(n).astype(np.float64) for n in names}


def test_adam_three_steps_beta_powers(gpu_required):
    """tf.train.AdamOptimizer (train.py:213): m, v slots and the beta-power bias correction over consecutive steps.  The
    oracle's adam_step is fed the ENGINE's own gradient of each step, so that only the optimiser arithmetic is compared:
    every trainable tensor, every element, fp32 rounding only (|dw| <= lr_t per step; bar 2e-6 * lr + fp32 ulp of w)."""
    B, N = 8, 128
    cfg = small_cfg(N=N)
    cfg["training"]["batch_size"] = B
    cfg["data"]["ntrain"] = 10 * B
    spec, P32 = oracle_params(cfg, seed=3)
    eng = alignnet3d.Engine(cfg)
    eng.set_variables(P32)
    names = R.trainable_names(spec)
    w = {n: P32[n].astype(np.float64).reshape(-1) for n in names}
    m = {n: np.zeros_like(w[n]) for n in names}
    v = {n: np.zeros_like(w[n]) for n in names}
    worst = 0.0
    for t in (1, 2, 3):
        d, u = _batch(cfg, B, N, 100 + t)
        lr = eng.state()["learning_rate"]
        eng.train_forward_backward(d["pcs1"], d["pcs2"], d, u)
        g = _all(eng, names, eng.get_gradient)
        eng.apply_gradients(1.0)
        assert eng.state()["step"] == t
        for n in names:
            w[n], m[n], v[n] = R.adam_step(w[n], g[n].reshape(-1), m[n], v[n], t, lr)
            got = eng.get_variable(n).astype(np.float64).reshape(-1)
            err = np.abs(got - w[n])
            worst = max(worst, float(err.max()))
            assert np.all(err <= 2e-6 * lr + 2.0 ** -22 * np.abs(w[n]) + 1e-9), (n, t, float(err.max()))
            w[n] = got   # continue from the engine's fp32 state (the slots stay the oracle's)
    print("adam, 3 steps: worst |w_engine - w_oracle| =", worst)
    eng.close()


def test_momentum_optimizer(gpu_required):
    """tf.train.MomentumOptimizer (train.py:211-212): accum = momentum * accum + g; w -= lr * accum, over three steps."""
    B, N = 8, 128
    cfg = small_cfg(N=N)
    cfg["training"]["batch_size"] = B
    cfg["training"]["optimizer"] = {"optimizer": "momentum", "momentum": 0.8}
    cfg["training"]["learning_rate"] = 0.01
    spec, P32 = oracle_params(cfg, seed=4)
    eng = alignnet3d.Engine(cfg)
    eng.set_variables(P32)
    names = R.trainable_names(spec)
    w = {n: P32[n].astype(np.float64).reshape(-1) for n in names}
    acc = {n: np.zeros_like(w[n]) for n in names}
    for t in (1, 2, 3):
        d, u = _batch(cfg, B, N, 200 + t)
        lr = eng.state()["learning_rate"]
        assert abs(lr - 0.01) < 1e-9
        eng.train_forward_backward(d["pcs1"], d["pcs2"], d, u)
        g = _all(eng, names, eng.get_gradient)
        eng.apply_gradients(1.0)
        for n in names:
            acc[n] = 0.8 * acc[n] + g[n].reshape(-1)
            ref = w[n] - lr * acc[n]
            got = eng.get_variable(n).astype(np.float64).reshape(-1)
            assert np.all(np.abs(got - ref) <= 1e-6 * lr * (1 + np.abs(acc[n])) + 2.0 ** -22 * np.abs(ref) + 1e-9), (n, t)
            w[n] = got
    eng.close()


def test_grad_scale_is_applied(gpu_required):
    """alignnet_apply_gradients(scale) = the 1/world of data-parallel training: momentum with scale 0.25 moves a quarter as far."""
    B, N = 8, 128
    cfg = small_cfg(N=N)
    cfg["training"]["batch_size"] = B
    cfg["training"]["optimizer"] = {"optimizer": "momentum", "momentum": 0.0}
    spec, P32 = oracle_params(cfg, seed=4)
    d, u = _batch(cfg, B, N, 7)
    moved = []
    for scale in (1.0, 0.25):
        eng = alignnet3d.Engine(cfg)
        eng.set_variables(P32)
        eng.train_forward_backward(d["pcs1"], d["pcs2"], d, u)
        eng.apply_gradients(scale)
        moved.append(eng.get_variable("fc1/weights").astype(np.float64) - P32["fc1/weights"])
        eng.close()
    np.testing.assert_allclose(moved[1], 0.25 * moved[0], rtol=1e-3, atol=3e-8)   # atol: one float32 ulp of the weights themselves (|w| < 0.25)


@pytest.mark.parametrize("per", ["epoch", "step"])
def test_schedules_as_the_engine_computes_them(gpu_required, per):
    """alignnet_set_step(k) + alignnet_get_state vs the oracle's learning_rate / bn_decay_schedule (train.py:133-174) at the
    staircase edges, at the 1e-5 floor and at the clip."""
    cfg = small_cfg()
    B, ntrain = 128, 20000 + 57   # ntrain not a multiple of B: nb_batches_per_epoch floors (train.py:142)
    cfg["training"]["batch_size"] = B
    cfg["data"]["ntrain"] = ntrain
    cfg["training"]["learning_rate"] = 0.005
    lr_step, bn_step = (30, 20) if per == "epoch" else (30 * 156 * B, 20 * 156 * B)
    cfg["training"]["lr_extension"] = {"mode": "decay", "per": per, "step": lr_step, "rate": 0.5}
    cfg["training"]["bn_extension"] = {"mode": "decay", "per": per, "step": bn_step, "rate": 0.5, "init": 0.5, "clip": 0.99}
    eng = alignnet3d.Engine(cfg)
    spe = ntrain // B
    edges = [0, 1, 20 * spe - 1, 20 * spe, 20 * spe + 1, 30 * spe - 1, 30 * spe, 60 * spe, 90 * spe - 1, 90 * spe, 8 * 30 * spe, 9 * 30 * spe,
             10 * 30 * spe, 7 * 20 * spe - 1, 7 * 20 * spe, 10 ** 7, 10 ** 9]
    seen_floor = seen_clip = False
    for k in edges:
        eng.set_step(k)
        st = eng.state()
        lr = R.learning_rate(k, B, ntrain, 0.005, lr_step, 0.5, per)
        bd = R.bn_decay_schedule(k, B, ntrain, 0.5, bn_step, 0.5, 0.99, per)
        assert st["step"] == k
        assert abs(st["learning_rate"] - lr) <= 1e-7 * lr, (k, st["learning_rate"], lr)
        assert abs(st["bn_decay"] - bd) <= 1e-6, (k, st["bn_decay"], bd)
        seen_floor |= lr == 1e-5
        seen_clip |= bd == 0.99
    assert seen_floor and seen_clip
    assert R.learning_rate(30 * spe - 1, B, ntrain, 0.005, lr_step, 0.5, per) == 0.005   # the edge is where the oracle says it is
    assert R.learning_rate(30 * spe, B, ntrain, 0.005, lr_step, 0.5, per) == 0.0025
    eng.close()


def test_step_result_reports_pre_update_schedule(gpu_required):
    """train.py:368 fetches learning_rate / bn_decay evaluated at the pre-increment `batch`; `step` is the value after."""
    B, N = 8, 64
    cfg = small_cfg(N=N)
    cfg["training"]["batch_size"] = B
    cfg["data"]["ntrain"] = 2 * B
    cfg["training"]["lr_extension"] = {"mode": "decay", "per": "epoch", "step": 1, "rate": 0.5}
    eng = alignnet3d.Engine(cfg, seed=1)
    d, u = _batch(cfg, B, N, 1)
    lrs = [eng.train_step(d["pcs1"], d["pcs2"], d, u) for _ in range(5)]
    assert [r["step"] for r in lrs] == [1, 2, 3, 4, 5]
    exp = [R.learning_rate(k, B, 2 * B, 0.005, 1, 0.5) for k in range(5)]
    np.testing.assert_allclose([r["learning_rate"] for r in lrs], exp, rtol=1e-6)
    eng.close()


def test_xavier_init(gpu_required):
    """alignnet_init_params vs utils/tf_util.py:10-49: weights uniform(-limit, limit) with limit = sqrt(6 / (fan_in + fan_out)),
    fans including the kernel extent (the first conv is [1,3] over one input channel: fan_in 3, fan_out 3*C), biases 0;
    :470-480: beta 0, gamma 1, EMA shadows 0 (TF zero slots).  SynthCars widths so that every tensor has >= 192 entries."""
    cfg = alignnet3d.default_model_config()
    spec = R.NetSpec.from_cfg(cfg)
    eng = alignnet3d.Engine(cfg, seed=11)
    eng2 = alignnet3d.Engine(cfg, seed=12)
    seen_first = 0
    for L in R.layer_table(spec):
        base = f"siamese/{L.name}" if L.siamese else L.name
        w = eng.get_variable(base + "/weights").astype(np.float64)
        limit = math.sqrt(6.0 / (L.fan_in + L.fan_out))
        if L.name.endswith("conv1"):
            assert (L.fan_in, L.fan_out) == (3, 3 * L.cout), (L.name, L.fan_in, L.fan_out)   # kh*kw*cin, kh*kw*cout with a [1,3] kernel
            seen_first += 1
        assert w.shape == (L.cin, L.cout) or w.size == L.cin * L.cout
        assert np.abs(w).max() <= limit * (1 + 1e-6), (L.name, np.abs(w).max(), limit)
        n = w.size
        # uniform(-a, a): variance a^2/3 (relative std of the sample variance = sqrt(0.8/n)), mean 0 +- a/sqrt(3n), and it fills the range
        assert abs(w.var() / (limit ** 2 / 3) - 1) < 5 * math.sqrt(0.8 / n) + 1e-3, (L.name, w.var(), limit ** 2 / 3)
        assert abs(w.mean()) < 5 * limit / math.sqrt(3 * n), (L.name, w.mean())
        assert np.abs(w).max() > limit * (1 - 20.0 / n), L.name
        assert not np.array_equal(w, eng2.get_variable(base + "/weights"))   # the seed matters
        assert np.all(eng.get_variable(base + "/biases") == 0)
    assert seen_first == 3
    for name, shp, trainable in eng.variables():
        v = eng.get_variable(name)
        if name.endswith("/gamma"):
            assert np.all(v == 1) and trainable
        elif name.endswith(("/beta", "/moving_mean", "/moving_var")):
            assert np.all(v == 0), name
            assert trainable == name.endswith("/beta")
    assert sorted(n for n, _, _ in eng.variables()) == sorted(n for n, _ in R.param_names(spec))
    ntrain = sum(s[0] * s[1] for _, s, t in eng.variables() if t)
    assert ntrain == 2_165_073
    eng.close(); eng2.close()


def test_device_dropout_stream(gpu_required):
    """tf.nn.dropout (utils/tf_util.py:571-573): x / keep * floor(keep + U[0,1)).  The device-side stream cannot reproduce TF's
    random_uniform; its distribution must: (i) the uniforms the kernels draw (read back through the debug export) are U[0,1) --
    chi-square over 64 bins, keep-rate at keep 0.7 within 4 sigma per head, no correlation between towers / heads / steps;
    (ii) the export IS the stream: a step that draws on the device equals, bit for bit, the step fed those uniforms explicitly."""
    B, N = 64, 64
    cfg = small_cfg(N=N)
    cfg["training"]["batch_size"] = B
    spec, P32 = oracle_params(cfg, seed=2)
    d = R.synth_pairs(B, N, seed=2, dtype=np.float32)
    eng = alignnet3d.Engine(cfg, seed=5)
    eng.set_variables(P32)
    u0 = eng.debug_dropout_uniforms(B)
    allu = np.concatenate([x.ravel() for x in u0])
    assert allu.min() >= 0.0 and allu.max() < 1.0
    n = allu.size
    hist = np.bincount((allu * 64).astype(int), minlength=64)
    chi2 = float(((hist - n / 64) ** 2 / (n / 64)).sum())
    assert chi2 < 63 + 5 * math.sqrt(2 * 63), chi2   # chi-square with 63 dof: mean 63, sd 11.2
    for x in u0:
        keep = float((np.floor(0.7 + x) == 1).mean())
        assert abs(keep - 0.7) < 4 * math.sqrt(0.21 / x.size), keep
    for a in range(5):
        for b in range(a + 1, 5):
            m = min(u0[a].size, u0[b].size)
            r = np.corrcoef(u0[a].ravel()[:m], u0[b].ravel()[:m])[0, 1]
            assert abs(r) < 5 / math.sqrt(m), (a, b, r)
    res_dev = eng.train_forward_backward(d["pcs1"], d["pcs2"], d, None)
    g_dev = eng.get_gradient("fc2/weights")
    eng.set_variables(P32)   # the EMA shadows moved
    res_exp = eng.train_forward_backward(d["pcs1"], d["pcs2"], d, u0)
    assert res_dev["loss"] == res_exp["loss"]
    np.testing.assert_array_equal(res_dev["pred_translations"], res_exp["pred_translations"])
    np.testing.assert_array_equal(g_dev, eng.get_gradient("fc2/weights"))
    eng.apply_gradients(1.0)   # step 0 -> 1: a new draw
    u1 = eng.debug_dropout_uniforms(B)
    for a, b in zip(u0, u1):
        assert not np.array_equal(a, b)
        assert abs(np.corrcoef(a.ravel(), b.ravel())[0, 1]) < 5 / math.sqrt(a.size)
    eng2 = alignnet3d.Engine(cfg, seed=6)
    assert not np.array_equal(eng2.debug_dropout_uniforms(B)[0], u0[0])   # cfg.seed matters
    eng.close(); eng2.close()
