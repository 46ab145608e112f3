"""GPU: BASELINE.json's full inference size (SynthCars widths, B = 256, N = 1024), where the fp64 oracle is too slow
to check every pair: size-independent properties of the path plus an oracle spot check on a subset."""
import numpy as np
import pytest

import alignnet3d
from oracle import alignnet_ref as R
from tests.helpers import oracle_params, compare_forward

pytestmark = pytest.mark.gpu
B, N = 256, 1024


@pytest.fixture(scope="module")
def setup():
    cfg = alignnet3d.default_model_config()
    spec, P32 = oracle_params(cfg)
    eng = alignnet3d.Engine(cfg)
    eng.set_variables(P32)
    d = R.synth_pairs(B, N, seed=4321, dtype=np.float32)
    base = eng.forward(d["pcs1"], d["pcs2"])
    yield cfg, spec, P32, eng, d, base
    eng.close()


def _stable(base, nb, eps=1e-3):
    def margin(lg):
        s = np.sort(lg[:, :nb], axis=1)
        return s[:, -1] - s[:, -2]
    return (margin(base["pred_pc1angle_logits"]) > eps) & (margin(base["pred_pc2angle_logits"]) > eps)


def test_finite_and_deterministic(gpu_required, setup):
    cfg, spec, P32, eng, d, base = setup
    again = eng.forward(d["pcs1"], d["pcs2"])
    for k, v in base.items():
        assert np.all(np.isfinite(v)), k
        np.testing.assert_array_equal(v, again[k])


def test_point_permutation_invariance(gpu_required, setup):
    """max-pool and mean are order-free (models/tp8.py:58,104); only the fp32 centroid summation order changes."""
    cfg, spec, P32, eng, d, base = setup
    perm = np.random.default_rng(0).permutation(N)
    out = eng.forward(d["pcs1"][:, perm], d["pcs2"][:, perm])
    ok = _stable(base, spec.num_bins)
    for k in base:
        sel = ok if k in ("pred_translations", "pred_remaining_angle_logits") else slice(None)
        np.testing.assert_allclose(out[k][sel], base[k][sel], rtol=1e-4, atol=1e-4, err_msg=k)


def test_translation_equivariance(gpu_required, setup):
    """Shifting both clouds by t shifts every predicted centre by t and leaves translation / logits unchanged."""
    cfg, spec, P32, eng, d, base = setup
    t = np.array([1.5, -2.25, 0.5], np.float32)
    out = eng.forward(d["pcs1"] + t, d["pcs2"] + t)
    ok = _stable(base, spec.num_bins)
    for k in ("pred_s1_pc1centers", "pred_s1_pc2centers", "pred_s2_pc1centers", "pred_s2_pc2centers"):
        np.testing.assert_allclose(out[k], base[k] + t, rtol=0, atol=2e-4, err_msg=k)
    for k in ("pred_pc1angle_logits", "pred_pc2angle_logits"):
        np.testing.assert_allclose(out[k], base[k], rtol=1e-4, atol=1e-4, err_msg=k)
    np.testing.assert_allclose(out["pred_translations"][ok], base["pred_translations"][ok], rtol=0, atol=3e-4)


def test_pairs_are_independent_and_towers_swap(gpu_required, setup):
    """Eval-mode pairs do not interact (any split of the batch gives the same rows); feeding (pcs2, pcs1) through
    towers with swapped BN sets swaps the per-tower outputs."""
    cfg, spec, P32, eng, d, base = setup
    part = eng.forward(d["pcs1"][100:137], d["pcs2"][100:137])
    for k in base:
        np.testing.assert_array_equal(part[k], base[k][100:137])
    swapped = {}
    for k, v in P32.items():
        k2 = k.replace("siamese_1/", "@@/").replace("siamese/", "siamese_1/").replace("@@/", "siamese/") if "/bn/" in k else k
        swapped[k2] = v
    eng2 = alignnet3d.Engine(cfg)
    eng2.set_variables(swapped)
    out = eng2.forward(d["pcs2"], d["pcs1"])
    eng2.close()
    np.testing.assert_array_equal(out["pred_s2_pc1centers"], base["pred_s2_pc2centers"])
    np.testing.assert_array_equal(out["pred_pc2angle_logits"], base["pred_pc1angle_logits"])


def test_oracle_spot_check(gpu_required, setup):
    cfg, spec, P32, eng, d, base = setup
    idx = np.arange(0, B, 32)
    P64 = {k: v.astype(np.float64) for k, v in P32.items()}
    ref, _, _ = R.get_model(P64, spec, d["pcs1"][idx].astype(np.float64), d["pcs2"][idx].astype(np.float64))
    worst, unstable = compare_forward({k: v[idx] for k, v in base.items()}, ref, spec.num_bins)
    print("full-size spot check: worst abs err", max(worst.values()), "unstable", unstable)


def test_training_reduces_loss_at_full_size(gpu_required):
    """A few Adam steps on one fixed batch of 256 pairs must lower the reference loss (train.py:368 semantics)."""
    cfg = alignnet3d.default_model_config()
    cfg["training"]["batch_size"] = B
    cfg["data"]["ntrain"] = 100 * B
    eng = alignnet3d.Engine(cfg, seed=3)
    d = R.synth_pairs(B, N, seed=7, dtype=np.float32)
    u = [np.full((B, 256), 0.9, np.float32)] * 5   # dropout off (keep + u >= 1) so that the loss is comparable step to step
    losses = [eng.train_step(d["pcs1"], d["pcs2"], d, u)["loss"] for _ in range(25)]
    print("losses", [round(x, 4) for x in losses])
    assert np.all(np.isfinite(losses)) and min(losses[-5:]) < 0.85 * losses[0]
    assert eng.state()["step"] == 25
    eng.close()


def test_bf16_training_learns_like_fp32(gpu_required):
    """BASELINE.json configs[2]: with the 128 -> C3 lifts on bf16 MFMA the network must still learn.  120 Adam steps on fresh
    synthetic batches (64 pairs x 512 points, SynthCars widths) from the same initialisation and the same batch sequence:
    both runs must cut the training loss by >= 30 % (mean of the first vs the last 10 steps), end within 15 % of each other,
    and give finite eval-mode predictions whose held-out translation errors are within a factor of two of each other (the two
    trajectories diverge step by step, so only the trend is comparable)."""
    Bs, Ns, steps = 64, 512, 120
    cfg = alignnet3d.default_model_config()
    cfg["model"]["num_points"] = Ns
    cfg["training"]["batch_size"] = Bs
    cfg["data"]["ntrain"] = 50 * Bs
    held = R.synth_pairs(Bs, Ns, seed=999, dtype=np.float32)
    final = {}
    for mode in (0, 1):
        eng = alignnet3d.Engine(cfg, seed=3)
        eng.set_option("train_matmul_bf16", mode)
        losses = []
        for k in range(steps):
            d = R.synth_pairs(Bs, Ns, seed=1000 + k, dtype=np.float32)
            losses.append(eng.train_step(d["pcs1"], d["pcs2"], d)["loss"])
        pred = eng.forward(held["pcs1"], held["pcs2"])["pred_translations"]
        err = float(np.linalg.norm(pred - held["translations"], axis=1).mean())
        final[mode] = (float(np.mean(losses[:10])), float(np.mean(losses[-10:])), err)
        eng.close()
        assert np.all(np.isfinite(losses)) and np.isfinite(pred).all()
    print("mean loss first / last 10 steps, held-out translation error: fp32", final[0], "bf16", final[1])
    for mode in (0, 1):
        assert final[mode][1] < 0.7 * final[mode][0], final
    assert abs(final[1][1] - final[0][1]) < 0.15 * final[0][1], final
    # held-out error after only 120 steps is a noisy statistic of a chaotic trajectory: the fp32 run alone moved from 0.62 to 0.39 when
    # the optimiser's constants changed in the 7th digit (round 2).  Bound: within a factor of two of each other.
    assert final[1][2] < 2.0 * final[0][2] + 0.05 and final[0][2] < 2.0 * final[1][2] + 0.05, final


def test_dgcnn_training_learns(gpu_required):
    """The DGCNN branch (tp8.py:30-46) trains end to end: 120 Adam steps on fresh synthetic batches (32 pairs x 256 points,
    SynthCars widths, k = 20 graph rebuilt every step) cut the training loss by >= 25 % (mean of the first vs the last 10
    steps), and the eval-mode forward afterwards (EMA statistics, same kNN kernels) is finite."""
    Bs, Ns, steps = 32, 256, 120
    cfg = alignnet3d.default_model_config()
    cfg["model"]["num_points"] = Ns
    cfg["model"]["backbone"] = "dgcnn"
    cfg["training"]["batch_size"] = Bs
    cfg["data"]["ntrain"] = 50 * Bs
    eng = alignnet3d.Engine(cfg, seed=3)
    losses = []
    for k in range(steps):
        d = R.synth_pairs(Bs, Ns, seed=2000 + k, dtype=np.float32)
        losses.append(eng.train_step(d["pcs1"], d["pcs2"], d)["loss"])
    held = R.synth_pairs(Bs, Ns, seed=998, dtype=np.float32)
    pred = eng.forward(held["pcs1"], held["pcs2"])["pred_translations"]
    eng.close()
    first, last = float(np.mean(losses[:10])), float(np.mean(losses[-10:]))
    print("dgcnn mean loss first / last 10 steps:", first, last)
    assert np.all(np.isfinite(losses)) and np.isfinite(pred).all()
    assert last < 0.75 * first, (first, last)


def _train_setup(backbone="pointnet", Bt=B, Nt=N, seed=5):
    cfg = alignnet3d.default_model_config()
    cfg["model"]["num_points"] = Nt
    cfg["model"]["backbone"] = backbone
    cfg["training"]["batch_size"] = Bt
    spec, P32 = oracle_params(cfg, seed=seed)
    d = R.synth_pairs(Bt, Nt, seed=seed, dtype=np.float32)
    rng = np.random.default_rng(seed)
    du = {k: rng.uniform(size=(Bt, 256)).astype(np.float32) for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")}
    return cfg, spec, P32, d, du


def _check_train_against_autograd(cfg, spec, P32, d, du, expect_kernel, pred_tol, loss_tol, ema_tol, cos_bar, rl2_bar, tensor_bar, fp32_context=True):
    """Train-mode forward (batch statistics over all 2 x B x N points, EMA), loss and every parameter gradient against the
    torch-autograd oracle in fp64 (backbones recomputed in the backward: oracle/alignnet_torch.py `checkpoint`).
    Every bar is 2 x the value this test printed at this round's head (profiles/r04_gpu_tests_fullsize.log holds the `full size:` lines):
    whole-gradient cosine / relative L2, the worst tensor (error relative to the tensor's own largest entry), predictions, loss, EMA.
    Why the gradient sits at 1e-2 and not at rounding level: at 256 x 1024 points the network takes 524 k max-pool decisions over 1024
    candidates each and normalises 256-row batches of nearly equal pooled features; the SAME oracle evaluated in fp32 is 2e-2 .. 6e-2
    (worst tensor) away from its own fp64 evaluation (printed for context, not used as a bar any more), and the engine's own gradient
    moves by 1.4e-2 when its inputs move by one ulp (tests/test_loopback_gpu.py prints that floor).  The small shapes of
    tests/test_train_gpu.py (5e-4 per tensor) are what catches a 1 % gradient bug; this is the same arithmetic at BASELINE.json's sizes."""
    from tests import test_train_gpu as TT
    us = [du[k] for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")]
    eng = alignnet3d.Engine(cfg)
    eng.set_variables(P32)
    ep_ref, loss_ref, grads, ema_ref = TT._oracle(cfg, P32, d, du, eng.state()["bn_decay"], checkpoint=True)
    if fp32_context:
        ep32, _, g32, _ = TT._oracle(cfg, P32, d, du, eng.state()["bn_decay"], checkpoint=True, dt=np.float32)
        gs = max(float(np.abs(v).max()) for v in grads.values())
        skip = {(f"siamese/{L.name}" if L.siamese else L.name) + "/biases" for L in R.layer_table(spec) if L.bn}
        rel32 = max(float(np.abs(g32[k].astype(np.float64) - grads[k]).max()) / (float(np.abs(grads[k]).max()) + 1e-5 * gs)
                    for k in grads if k not in skip)
        pred32 = max(float(np.abs(ep32[k] - ep_ref[k]).max()) for k in ep_ref)
        print("fp32 oracle vs fp64 oracle: worst relative gradient error %.2e, worst prediction error %.2e" % (rel32, pred32))
    res = eng.train_forward_backward(d["pcs1"], d["pcs2"], d, us)
    assert eng.get_option("last_train_kernel") == expect_kernel
    worst_pred = max(float(np.abs(res[k] - ep_ref[k]).max()) for k in ep_ref)
    worst_ema = 0.0
    ema_fail = []
    for k, v in ema_ref.items():
        got = eng.get_variable(k)
        if not np.allclose(got, v, rtol=ema_tol, atol=0.1 * ema_tol):
            ema_fail.append(k)
        worst_ema = max(worst_ema, float(np.abs(got - v).max()))
    gscale = max(float(np.abs(v).max()) for v in grads.values())
    bn_bias = {(f"siamese/{L.name}" if L.siamese else L.name) + "/biases" for L in R.layer_table(spec) if L.bn}
    bad, rel = {}, {}
    for name in R.trainable_names(spec):
        g = eng.get_gradient(name).astype(np.float64)
        ref = grads[name].reshape(g.shape)
        if name in bn_bias:
            assert np.abs(g).max() == 0.0 and np.abs(ref).max() < 1e-9 * gscale, name
            continue
        err = float(np.abs(g - ref).max())
        if np.abs(ref).max() > 1e-6 * gscale:
            rel[name] = err / float(np.abs(ref).max())
        if err > tensor_bar * float(np.abs(ref).max()) + 1e-5 * gscale:
            bad[name] = (err, float(np.abs(ref).max()))
    names = [n for n in R.trainable_names(spec) if n not in bn_bias]
    gv = np.concatenate([eng.get_gradient(n).astype(np.float64).ravel() for n in names])
    rv = np.concatenate([np.asarray(grads[n], np.float64).ravel() for n in names])
    cos = float(gv @ rv / (np.linalg.norm(gv) * np.linalg.norm(rv)))
    rl2 = float(np.linalg.norm(gv - rv) / np.linalg.norm(rv))
    eng.close()
    print("full size: loss %.6f (oracle %.6f), worst prediction err %.2e, worst EMA err %.2e, whole gradient: cosine %.6f, relative L2 error %.2e, "
          "worst relative gradient errors %s" % (res["loss"], loss_ref, worst_pred, worst_ema, cos, rl2, sorted(rel.items(), key=lambda kv: -kv[1])[:4]))
    assert worst_pred <= pred_tol, worst_pred
    assert abs(res["loss"] - loss_ref) <= loss_tol * max(1.0, abs(loss_ref)), (res["loss"], loss_ref)
    assert not ema_fail, ema_fail
    assert cos >= cos_bar and rl2 <= rl2_bar, (cos, rl2)
    assert not bad, bad


def test_train_fp32_full_size_matches_autograd(gpu_required):
    """BASELINE.json configs[2]'s shape in fp32: SynthCars widths, 256 pairs x 1024 points -- the kernel instantiations with the
    widths compiled in, whole-cloud tile walks, 512 workgroups, the B x B loss terms at B = 256."""
    cfg, spec, P32, d, du = _train_setup()
    # measured: predictions 1.3e-4, loss 1e-6, EMA 2.5e-6, cosine 0.999911, relative L2 1.3e-2, worst tensor fc1/weights 6.3e-2
    _check_train_against_autograd(cfg, spec, P32, d, du, expect_kernel=1, pred_tol=2.5e-4, loss_tol=1e-5, ema_tol=5e-5, cos_bar=0.9998, rl2_bar=2.7e-2, tensor_bar=0.125)


def test_train_bf16_full_size_matches_rounded_oracle(gpu_required):
    """BASELINE.json configs[2] at its own size: bf16 MFMA convs, 256 pairs x 1024 points, against the oracle that models the
    operand rounding (same criteria as tests/test_train_gpu.py::test_bf16_lift_matches_rounded_oracle)."""
    from tests import test_train_gpu as TT
    cfg, spec, P32, d, du = _train_setup()
    TT.bf16_check(cfg, spec, P32, d, du, B, expect_kernel=3, checkpoint=True)


def test_train_dgcnn_n1024_matches_autograd(gpu_required):
    """DGCNN training at N = 1024 (the kNN kernel's 16-slot instantiation at its limit, 16 tiles per cloud, 20 neighbour slots,
    SynthCars widths -> dg_train_fwd<64> / dg_train_bwd_edge<64, 128>): B = 4 keeps the [B*N*k, C] autograd oracle in memory."""
    cfg, spec, P32, d, du = _train_setup("dgcnn", Bt=4, Nt=1024, seed=7)
    # (two max-pools and 4-row batch statistics in the heads: the fp32 evaluation of the oracle itself is 3.8e-2 from its fp64 one here)
    # (EMA bound 2e-4: with the point conv on 128-point tiles one of the 512 fc1 moving means -- four-row batch statistics -- sits 1.4e-5 from the
    #  fp64 value at |v| = 0.03, just outside 1e-4 |v| + 1e-5)
    # measured: predictions 2.3e-4, loss 2e-6, EMA 4.1e-5, cosine 0.999965, relative L2 9.1e-3, worst tensor siamese_1/embedding/conv1/bn/beta 3.8e-2
    _check_train_against_autograd(cfg, spec, P32, d, du, expect_kernel=5, pred_tol=5e-4, loss_tol=1e-5, ema_tol=2e-4, cos_bar=0.9999, rl2_bar=1.9e-2, tensor_bar=8e-2)


def test_train_b2048_matches_autograd(gpu_required):
    """BASELINE.json configs[3]'s arithmetic on ONE GPU: the global batch of 2048 pairs (KITTITrackletsCarsPersonsHard: SynthCars
    widths) in a single step -- the [B, B] loss terms of models/tp8.py:279,327 at 4 M entries, the whole-batch tf.cond (:288), the
    4096-row head BatchNorms, 4096 workgroups per backbone launch.  N = 128 keeps the fp64 autograd oracle at the cost of the
    256 x 1024 test (the same 524 k points)."""
    cfg, spec, P32, d, du = _train_setup(Bt=2048, Nt=128, seed=11)
    # (2048-row batch statistics in the heads: the fp32 evaluation of the oracle itself is 4e-2 from its fp64 one on the smallest gradients)
    # measured: predictions 1.5e-4, loss 1e-6, EMA 8e-7, cosine 0.999979, relative L2 6.4e-3, worst tensor fc1/bn/beta 3.3e-2
    _check_train_against_autograd(cfg, spec, P32, d, du, expect_kernel=1, pred_tol=3e-4, loss_tol=1e-5, ema_tol=2e-5, cos_bar=0.99995, rl2_bar=1.3e-2, tensor_bar=7e-2)


def test_train_dgcnn_n4096_matches_autograd(gpu_required):
    """BASELINE.json configs[4]'s training half at its own N: DGCNN at N = 4096 (knn_kernel<64>, 64 tiles per cloud, SynthCars
    widths -> dg_train_fwd<64> / dg_train_bwd_edge<64, 128>) against fp64 autograd; B = 4 (655 k edge rows, [2B, N, N] distance
    matrices in the oracle; two-row batch statistics in the heads are singular, so not B = 2)."""
    cfg, spec, P32, d, du = _train_setup("dgcnn", Bt=4, Nt=4096, seed=9)
    # The fp32 kNN graph differs from the fp64 oracle's wherever the 20th and 21st neighbour of a query are closer than fp32 rounding of the
    # distance expression (N = 4096: dozens of queries per cloud), and four-row batch statistics in the heads amplify one changed neighbour
    # into per cents on the smallest tensors (siamese/embedding/conv3/bn/beta: 0.42 of its 1e-3-sized entries): the per-tensor bar only
    # bounds that tensor, the whole-gradient bars carry the comparison.
    # measured: predictions 4.4e-3 (the fp32 oracle's own: 7.7e-3), loss 2.2e-4, EMA 2.3e-3, cosine 0.999856, relative L2 1.7e-2
    _check_train_against_autograd(cfg, spec, P32, d, du, expect_kernel=5, pred_tol=9e-3, loss_tol=5e-4, ema_tol=5e-3, cos_bar=0.9997, rl2_bar=3.4e-2, tensor_bar=0.85)


def test_train_dgcnn_n4096_b512_runs(gpu_required):
    """configs[4] at one GPU's share of the batch (4096 / 8 = 512 pairs, N = 4096): two full training steps (fp32, then bf16
    convs) -- finite loss, the expected kernel instantiations, a parameter update -- and an eval forward afterwards."""
    Bs, Ns = 512, 4096
    cfg = alignnet3d.default_model_config()
    cfg["model"]["num_points"] = Ns
    cfg["model"]["backbone"] = "dgcnn"
    cfg["training"]["batch_size"] = Bs
    eng = alignnet3d.Engine(cfg, seed=3)
    d = R.synth_pairs(Bs, Ns, seed=21, dtype=np.float32)
    w0 = eng.get_variable("siamese/embedding/conv3/weights").copy()
    r0 = eng.train_step(d["pcs1"], d["pcs2"], d)
    assert eng.get_option("last_train_kernel") == 5 and np.isfinite(r0["loss"])
    eng.set_option("train_matmul_bf16", 1)
    r1 = eng.train_step(d["pcs1"], d["pcs2"], d)
    assert eng.get_option("last_train_kernel") == 7 and np.isfinite(r1["loss"])
    w1 = eng.get_variable("siamese/embedding/conv3/weights")
    assert np.isfinite(w1).all() and np.abs(w1 - w0).max() > 0
    out = eng.forward(d["pcs1"][:8], d["pcs2"][:8])
    eng.close()
    assert all(np.isfinite(v).all() for v in out.values())
    print("dgcnn N=4096 B=512: losses", r0["loss"], r1["loss"])
