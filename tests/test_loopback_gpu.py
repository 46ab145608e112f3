"""The engine's multi-rank code with DIFFERENT shards on one GPU: W engines of this process joined by the in-process loopback
communicator (include/alignnet_hip.h: alignnet_comm_loopback_id; csrc/comm_loopback.h), one host thread per rank.

The reference is single-device (train.py:189): its BatchNorm moments are over the whole batch (utils/tf_util.py:474) and its loss
couples all samples ([B, B] broadcasts models/tp8.py:279,327; whole-batch tf.cond :288).  A data-parallel step with `sync_bn` +
`global_loss` must therefore BE the single-device step at the concatenated batch: summed gradient, loss, EMA shadows, predictions.
The checker is a single engine (no communicator) on the whole batch -- itself held to the oracle by tests/test_train_gpu.py and
tests/test_fullsize_gpu.py -- plus, at B = 2048, the fp64 autograd oracle directly (BASELINE.json configs[3])."""
import threading

import numpy as np
import pytest

import alignnet3d
from oracle import alignnet_ref as R
from tests.helpers import small_cfg, oracle_params

pytestmark = pytest.mark.gpu
LABELS = ("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles")
U = ("s1_0", "s2_0", "s1_1", "s2_1", "rem")
STD = dict(s1=(64, 128, 96), s2=(64, 128, 128), emb=(64, 128, 160))


def run_ranks(W, cfg, body, options=(), variables=None):
    """W ranks = W threads, each with its own Engine joined to one loopback group; returns [body(rank, engine)] or raises the
    first rank's exception (a failing rank breaks the group, so the others return with an error instead of waiting)."""
    uid = alignnet3d.Engine.comm_loopback_id()
    out, err = [None] * W, [None] * W

    def worker(r):
        eng = None
        try:
            eng = alignnet3d.Engine(cfg)
            if variables is not None:
                eng.set_variables(variables)
            for k, v in options:
                eng.set_option(k, v)
            eng.comm_init(r, W, uid)
            assert eng.get_option("comm_world") == W
            out[r] = body(r, eng)
        except BaseException as e:  # noqa: BLE001
            err[r] = e
        finally:
            if eng is not None:
                eng.close()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in th), "a rank thread is stuck"
    first = [e for e in err if e is not None and "rendezvous broken" not in str(e)] or [e for e in err if e is not None]
    if first:
        raise first[0]
    return out


def shard(d, du, r, W):
    B = d["pcs1"].shape[0]
    lo, hi = r * B // W, (r + 1) * B // W
    return {k: v[lo:hi] for k, v in d.items()}, [du[k][lo:hi] for k in U], lo, hi


def trainable(eng):
    return [n for n, _, t in eng.variables() if t]


def flat_grad(eng, names):
    return np.concatenate([eng.get_gradient(n).astype(np.float64).ravel() for n in names])


def single_engine(cfg, P32, d, du, options=()):
    eng = alignnet3d.Engine(cfg)
    eng.set_variables(P32)
    for k, v in options:
        eng.set_option(k, v)
    res = eng.train_forward_backward(d["pcs1"], d["pcs2"], d, [du[k] for k in U])
    names = trainable(eng)
    g = {n: eng.get_gradient(n).astype(np.float64) for n in names}
    ema = {n: eng.get_variable(n) for n, _, t in eng.variables() if not t}
    eng.close()
    return res, g, ema


def setup(backbone, N, B, std=True, seed=5, widths=None, fcw=32):
    w = widths or (STD if std else dict(s1=(32, 64, 96), s2=(32, 64, 128), emb=(32, 64, 160)))
    if backbone == "dgcnn" and not std and widths is None:
        w = dict(s1=(32, 64, 96), s2=(32, 64, 128), emb=(64, 128, 160))
    cfg = small_cfg(N=N, nb=12, fc=(64, fcw), backbone=backbone, **w)
    cfg["training"]["batch_size"] = B
    spec, P32 = oracle_params(cfg, seed=seed)
    d = R.synth_pairs(B, N, seed=seed, dtype=np.float32)
    rng = np.random.default_rng(seed)
    du = {k: rng.uniform(size=(B, fcw)).astype(np.float32) for k in U}
    return cfg, spec, P32, d, du


def sharded_step(W, cfg, P32, d, du, options):
    """sync_bn + global_loss data-parallel forward/backward of W ranks on distinct shards, then the gradient all-reduce."""
    def body(r, eng):
        sd, su, lo, hi = shard(d, du, r, W)
        res = eng.train_forward_backward(sd["pcs1"], sd["pcs2"], sd, su)
        names = trainable(eng)
        local = {n: eng.get_gradient(n).astype(np.float64) for n in names}
        eng.comm_allreduce_grads()
        eng.synchronize()
        summed = {n: eng.get_gradient(n).astype(np.float64) for n in names}
        ema = {n: eng.get_variable(n) for n, _, t in eng.variables() if not t}
        return dict(res=res, local=local, summed=summed, ema=ema, lo=lo, hi=hi)
    return run_ranks(W, cfg, body, options=tuple(options) + (("sync_bn", 1), ("global_loss", 1)), variables=P32)


def compare_with_single(ranks, single, gtol, ptol, ltol=2e-6, etol=2e-5, label=""):
    rf, gf, ef = single
    names = list(gf)
    gs = max(float(np.abs(v).max()) for v in gf.values())
    worst, worst_name = 0.0, ""
    for rk in ranks:
        assert abs(rk["res"]["loss"] - rf["loss"]) <= ltol * max(1.0, abs(rf["loss"])), (rk["res"]["loss"], rf["loss"])
        for i, s in enumerate(rk["res"]["summaries"].values()):
            assert abs(s - list(rf["summaries"].values())[i]) <= 10 * ltol * max(1.0, abs(s))
        for k in ("pred_translations", "pred_remaining_angle_logits", "pred_s1_pc1centers", "pred_s2_pc2centers", "pred_pc1angle_logits", "pred_pc2angle_logits"):
            np.testing.assert_allclose(rk["res"][k], rf[k][rk["lo"]:rk["hi"]], rtol=ptol, atol=ptol, err_msg=k)
        for n, v in rk["ema"].items():
            np.testing.assert_allclose(v, ef[n], rtol=etol, atol=etol, err_msg=n)
        for n in names:   # every rank holds the same sums, bit for bit (the loopback sums in rank order on every rank)
            np.testing.assert_array_equal(rk["summed"][n], ranks[0]["summed"][n], err_msg=n)
    # the all-reduced gradient is the sum of the ranks' local ones ...
    for n in names:
        tot = sum(rk["local"][n] for rk in ranks)
        np.testing.assert_allclose(ranks[0]["summed"][n], tot, rtol=1e-6, atol=1e-7 * gs, err_msg=n)
    # ... and equals the single engine's gradient at the concatenated batch
    for n in names:
        err = float(np.abs(ranks[0]["summed"][n] - gf[n]).max())
        rel = err / (float(np.abs(gf[n]).max()) + 1e-3 * gs)
        if rel > worst:
            worst, worst_name = rel, n
        assert err <= gtol * float(np.abs(gf[n]).max()) + 1e-2 * gtol * gs, (n, err, float(np.abs(gf[n]).max()))
    ga = np.concatenate([ranks[0]["summed"][n].ravel() for n in names]); gb = np.concatenate([gf[n].ravel() for n in names])
    cos = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))
    rl2 = float(np.linalg.norm(ga - gb) / np.linalg.norm(gb))
    print("loopback %s: %d ranks vs one engine at the concatenated batch: loss %.7f / %.7f, whole gradient cosine %.8f, relative L2 %.2e, "
          "worst tensor %s %.2e" % (label, len(ranks), ranks[0]["res"]["loss"], rf["loss"], cos, rl2, worst_name, worst))
    return cos, rl2


@pytest.mark.parametrize("W", [2, 8])
@pytest.mark.parametrize("backbone,bf16,std", [("pointnet", 0, False), ("pointnet", 0, True), ("pointnet", 1, True), ("dgcnn", 0, True), ("dgcnn", 1, True)])
def test_sharded_step_equals_single_engine(gpu_required, W, backbone, bf16, std):
    """W = 2 and 8 ranks with distinct shards of one batch of 16 pairs, every fused backbone mode (the modes of
    tests/test_train_gpu.py::test_sync_bn_*): loss, predictions, EMA shadows and the summed gradient equal one engine's step."""
    N, B = 128, 16
    cfg, spec, P32, d, du = setup(backbone, N, B, std=std, seed=7 if backbone == "dgcnn" else 5)
    opts = (("train_matmul_bf16", bf16),)
    single = single_engine(cfg, P32, d, du, opts)
    ranks = sharded_step(W, cfg, P32, d, du, opts)
    # fp32: the shards' partial sums are added in another order than the single engine's (measured <= 3e-5 of a tensor's largest entry);
    # bf16 operands: a value next to a rounding boundary of h2 / dy2 may round the other way (the same noise the rounded-oracle tests carry)
    compare_with_single(ranks, single, gtol=(3e-2 if bf16 else 5e-4), ptol=(2e-2 if bf16 else 2e-5), ltol=(2e-4 if bf16 else 2e-6),
                        etol=(2e-3 if bf16 else 2e-5), label="%s %s W=%d" % (backbone, "bf16" if bf16 else "fp32", W))


@pytest.mark.parametrize("backbone,tail", [("pointnet", 1), ("pointnet", 0), ("dgcnn", 1)])
def test_sharded_step_general_depth(gpu_required, backbone, tail):
    """Stages that train layer by layer (any depth / widths; gen_stat_finish's three launches around two all-reduces, gen_bn_bwd_finish's
    coefficient totals) with four distinct shards."""
    N, B, W = 128, 16, 4
    if backbone == "dgcnn":
        w = dict(s1=(32, 32, 64, 96), s2=(48, 96, 128), emb=(64, 160))
    else:
        w = dict(s1=(48, 96, 160), s2=(32, 32, 32, 64, 128), emb=(40, 72, 104))
    cfg, spec, P32, d, du = setup(backbone, N, B, widths=w)
    opts = (("train_fused_tail", tail),)
    single = single_engine(cfg, P32, d, du, opts)
    ranks = sharded_step(W, cfg, P32, d, du, opts)
    compare_with_single(ranks, single, gtol=5e-4, ptol=2e-5, label="%s general depth tail=%d W=%d" % (backbone, tail, W))


def test_local_bn_averaged_gradient_is_mean_of_shard_gradients(gpu_required):
    """Default data-parallel semantics ("local BN"): every rank is a reference run on its own shard; the step's all-reduce sums the
    gradients and the optimiser scales by 1 / world.  Four ranks: the all-reduced gradient equals the sum of four single-engine
    gradients on the four shards bit for bit up to summation order, in both all-reduce modes; after alignnet_train_step every rank
    holds identical parameters, equal to one engine applying the mean gradient."""
    N, B, W = 128, 16, 4
    cfg, spec, P32, d, du = setup("pointnet", N, B, std=True)
    cfg["training"]["batch_size"] = B // W
    per_shard = []
    for r in range(W):
        sd, su, lo, hi = shard(d, du, r, W)
        res, g, ema = single_engine(cfg, P32, sd, {k: su[i] for i, k in enumerate(U)})
        per_shard.append(g)
    names = list(per_shard[0])
    mean = {n: sum(g[n] for g in per_shard) / W for n in names}
    for overlap in (1, 0):
        def body(r, eng):
            sd, su, lo, hi = shard(d, du, r, W)
            eng.train_forward_backward(sd["pcs1"], sd["pcs2"], sd, su)
            local = {n: eng.get_gradient(n).astype(np.float64) for n in names}
            eng.set_variables(P32)   # the EMA shadows moved: same starting point for the real step
            res = eng.train_step(sd["pcs1"], sd["pcs2"], sd, su)
            return dict(local=local, w={n: eng.get_variable(n) for n in names}, buckets=eng.get_option("comm_buckets"), order=eng.get_option("comm_order"))
        ranks = run_ranks(W, cfg, body, options=(("allreduce_overlap", overlap),), variables=P32)
        for r in range(W):
            for n in names:
                np.testing.assert_array_equal(ranks[r]["local"][n], per_shard[r][n], err_msg=n)   # a rank IS a single engine on its shard
                np.testing.assert_array_equal(ranks[r]["w"][n], ranks[0]["w"][n], err_msg=n)      # identical parameters on every rank
            assert ranks[r]["buckets"] == (3 if overlap else 0)
            # bucketed: every stage's bucket is issued right behind that stage's backward, before the next stage's backward is queued
            assert ranks[r]["order"] == (362514 if overlap else 321), ranks[r]["order"]
        # one engine applying the mean gradient: Adam's first step moves every weight by lr * sign(g) where |g| is not tiny
        w0 = np.concatenate([np.asarray(P32[n], np.float32).ravel() for n in names])
        w1 = np.concatenate([ranks[0]["w"][n].ravel() for n in names])
        gm = np.concatenate([mean[n].ravel() for n in names])
        big = np.abs(gm) > 1e-3 * np.abs(gm).max()
        assert np.all(np.sign(w0[big] - w1[big]) == np.sign(gm[big]))
        step = np.abs(w0[big] - w1[big])
        assert np.allclose(step, 0.005, rtol=2e-3), (step.min(), step.max())


def test_collective_mismatch_and_failed_rank_do_not_hang(gpu_required):
    """A rank that fails (or issues another collective) breaks the group: the other ranks return an error, nobody waits forever."""
    N, B, W = 128, 8, 2
    cfg, spec, P32, d, du = setup("pointnet", N, B, std=False)

    def body(r, eng):
        sd, su, lo, hi = shard(d, du, r, W)
        if r == 1:
            eng.set_option("sync_bn", 0)   # rank 1 issues no per-layer sums: the first collectives of the two ranks differ
        eng.train_forward_backward(sd["pcs1"], sd["pcs2"], sd, su)
        eng.comm_allreduce_grads()
        return True
    with pytest.raises(alignnet3d.engine.EngineError) as e:
        run_ranks(W, cfg, body, options=(("sync_bn", 1),), variables=P32)
    assert "loopback communicator" in str(e.value)


def test_configs3_partition_8x256_n1024(gpu_required):
    """BASELINE.json configs[3] at its real partition on one GPU: 8 ranks x 256 pairs, N = 1024, SynthCars widths, sync_bn +
    global_loss, against ONE engine taking all 2048 pairs (4096-row head BatchNorms, the [B, B] loss terms at 4 M entries).  fp32 and
    bf16 convs.  Also the first execution of B = 2048 x N = 1024 on the single engine at all."""
    W, Bs, N = 8, 256, 1024
    cfg = alignnet3d.default_model_config()
    cfg["training"]["batch_size"] = W * Bs
    spec, P32 = oracle_params(cfg, seed=11)
    d = R.synth_pairs(W * Bs, N, seed=11, dtype=np.float32)
    rng = np.random.default_rng(11)
    du = {k: rng.uniform(size=(W * Bs, 256)).astype(np.float32) for k in U}
    for bf16 in (0, 1):
        opts = (("train_matmul_bf16", bf16),)
        single = single_engine(cfg, P32, d, du, opts)
        assert np.isfinite(single[0]["loss"])
        ranks = sharded_step(W, cfg, P32, d, du, opts)
        # full size: 2 M points per tower behind every batch statistic; the fp32 bars are 2 x what this test printed when it was
        # written (profiles/r04_gpu_tests_fullsize.log), the per-tensor bar only has to catch a wrong factor or a wrong row
        cos, rl2 = compare_with_single(ranks, single, gtol=(5e-2 if bf16 else 2e-3), ptol=(3e-2 if bf16 else 5e-5), ltol=(5e-4 if bf16 else 5e-6),
                                       etol=(5e-3 if bf16 else 5e-5), label="configs[3] 8 x 256 x 1024 %s" % ("bf16" if bf16 else "fp32"))
        assert cos > (0.9995 if bf16 else 0.999999) and rl2 < (3e-2 if bf16 else 1e-3), (cos, rl2)


def test_configs3_b2048_sharded_matches_autograd(gpu_required):
    """configs[3]'s arithmetic against the ORACLE through the sharded path: 8 ranks x 256 pairs at N = 128 (the shape of
    tests/test_fullsize_gpu.py::test_train_b2048_matches_autograd, whose fp64 autograd oracle fits) -- loss, EMA and the all-reduced
    gradient of the data-parallel step against torch autograd on the whole batch."""
    from tests import test_train_gpu as TT
    W, Bs, N = 8, 256, 128
    cfg = alignnet3d.default_model_config()
    cfg["model"]["num_points"] = N
    cfg["training"]["batch_size"] = W * Bs
    spec, P32 = oracle_params(cfg, seed=11)
    d = R.synth_pairs(W * Bs, N, seed=11, dtype=np.float32)
    rng = np.random.default_rng(11)
    du = {k: rng.uniform(size=(W * Bs, 256)).astype(np.float32) for k in U}
    ranks = sharded_step(W, cfg, P32, d, du, ())
    ep_ref, loss_ref, grads, ema_ref = TT._oracle(cfg, P32, d, du, 0.5, checkpoint=True)
    assert abs(ranks[0]["res"]["loss"] - loss_ref) <= 1e-4 * max(1.0, abs(loss_ref)), (ranks[0]["res"]["loss"], loss_ref)
    for rk in ranks:
        for k in ep_ref:
            np.testing.assert_allclose(rk["res"][k], ep_ref[k][rk["lo"]:rk["hi"]], rtol=2.5e-4, atol=2.5e-4, err_msg=k)
    for k, v in ema_ref.items():
        np.testing.assert_allclose(ranks[0]["ema"][k], v, rtol=1e-4, atol=1e-5, err_msg=k)
    bn_bias = {(f"siamese/{L.name}" if L.siamese else L.name) + "/biases" for L in R.layer_table(spec) if L.bn}
    names = [n for n in R.trainable_names(spec) if n not in bn_bias]
    gv = np.concatenate([ranks[0]["summed"][n].ravel() for n in names])
    rv = np.concatenate([np.asarray(grads[n], np.float64).ravel() for n in names])
    cos = float(gv @ rv / (np.linalg.norm(gv) * np.linalg.norm(rv)))
    rl2 = float(np.linalg.norm(gv - rv) / np.linalg.norm(rv))
    print("configs[3] sharded (8 x 256, N = 128) vs fp64 autograd: loss %.6f / %.6f, whole gradient cosine %.7f, relative L2 %.2e" % (ranks[0]["res"]["loss"], loss_ref, cos, rl2))
    assert cos > 0.99999 and rl2 < 5e-3, (cos, rl2)
