"""GPU parity: HIP eval-mode forward through the C ABI vs the fp64 oracle on the same
fp32-representable parameters and inputs.  Bar: 1e-4 (BASELINE.json north_star)."""
import numpy as np
import pytest

import alignnet3d
from oracle import alignnet_ref as R
from tests.helpers import small_cfg, oracle_params, compare_forward

pytestmark = pytest.mark.gpu


def _run(cfg, B, seed=1234, split=False, options=()):
    spec, P32 = oracle_params(cfg)
    eng = alignnet3d.Engine(cfg)
    eng.set_variables(P32)
    for k, v in options:
        eng.set_option(k, v)
        assert eng.get_option(k) == v
    if split:
        eng.set_option("infer_matmul_bf16x3", 1)
        assert eng.get_option("infer_matmul_bf16x3") == 1
    d = R.synth_pairs(B, spec.num_points, seed=seed, dtype=np.float32)
    ep = eng.forward(d["pcs1"], d["pcs2"])
    global LAST_KERNEL
    LAST_KERNEL = eng.last_backbone_kernel()   # which instantiation the embedding backbone (the last launch) ran
    P64 = {k: v.astype(np.float64) for k, v in P32.items()}
    ref, _, _ = R.get_model(P64, spec, d["pcs1"].astype(np.float64), d["pcs2"].astype(np.float64))
    eng.close()
    return ep, ref, spec


LAST_KERNEL = "none"


@pytest.mark.parametrize("N,B", [(128, 5), (100, 3), (256, 33), (37, 1)])
def test_forward_small_widths(gpu_required, N, B):
    cfg = small_cfg(N=N)
    ep, ref, spec = _run(cfg, B)
    worst, unstable = compare_forward(ep, ref, spec.num_bins)
    print("worst abs err", worst, "unstable pairs", unstable)
    assert unstable <= max(1, B // 4)


@pytest.mark.parametrize("tile,B,kernel", [(128, 8, "pointnet_fused<64,128,k16>"),          # the instantiation bench.py times (B = 256 picks it by itself)
                                           (0, 8, "pointnet_fused<64,128,k16,tp64>"),      # automatic at a serving-size batch: 64-point tiles
                                           (0, 40, "pointnet_fused<64,128,k16>")])         # automatic at 2B x 8 = 640 tiles of 128 points
def test_forward_synthcars_widths_n1024(gpu_required, tile, B, kernel):
    """The shipped widths at N = 1024 against the fp64 oracle, on both tile shapes of the fused backbone ("infer_tile_points": 0 = from the batch -- 64-point tiles
    while the 128-point tiling would cover at most half of the CUs)."""
    cfg = alignnet3d.default_model_config()
    ep, ref, spec = _run(cfg, B, options=(("infer_tile_points", tile),))
    worst, unstable = compare_forward(ep, ref, spec.num_bins)
    print("tile option", tile, "B", B, "worst abs err", worst, "unstable pairs", unstable)
    assert unstable <= max(2, B // 4)
    assert LAST_KERNEL == kernel, LAST_KERNEL


def test_forward_tile_shapes_are_bit_identical(gpu_required):
    """64- and 128-point tiles run the same MFMA k-order per output and a max over the same values: identical bits (so the automatic choice by batch size
    never changes a result), on a ragged cloud size and both batch regimes."""
    cfg = alignnet3d.default_model_config()
    cfg["model"]["num_points"] = 1000
    spec, P32 = oracle_params(cfg)
    for B in (3, 40):
        d = R.synth_pairs(B, 1000, seed=5, dtype=np.float32)
        outs = {}
        for tile in (64, 128, 0):
            eng = alignnet3d.Engine(cfg)
            eng.set_variables(P32)
            eng.set_option("infer_tile_points", tile)
            outs[tile] = eng.forward(d["pcs1"], d["pcs2"])
            eng.close()
        for tile in (128, 0):
            for k in outs[64]:
                np.testing.assert_array_equal(outs[tile][k], outs[64][k], err_msg="%s tile %d B %d" % (k, tile, B))


@pytest.mark.parametrize("N,B", [(128, 5), (100, 3), (256, 33), (37, 1)])
def test_forward_split_bf16_small_widths(gpu_required, N, B):
    """Opt-in split-bf16 backbone ("infer_matmul_bf16x3": x = hi + lo bf16, three bf16 MFMAs per product, fp32 accumulate):
    held to the SAME bar as the exact-fp32 path (1e-4, compare_forward)."""
    cfg = small_cfg(N=N)
    ep, ref, spec = _run(cfg, B, split=True)
    worst, unstable = compare_forward(ep, ref, spec.num_bins)
    print("split-bf16 worst abs err", worst, "unstable pairs", unstable)
    assert unstable <= max(1, B // 4)


def test_forward_split_bf16_synthcars_widths_n1024(gpu_required):
    cfg = alignnet3d.default_model_config()
    ep, ref, spec = _run(cfg, 8, split=True)
    worst, unstable = compare_forward(ep, ref, spec.num_bins)
    ep32, _, _ = _run(cfg, 8)
    print("split-bf16 worst abs err", worst, "vs exact-fp32 path", {k: float(np.abs(ep32[k] - ref[k]).max()) for k in ref})
    assert unstable <= 2
    assert any(not np.array_equal(ep[k], ep32[k]) for k in ep), "option had no effect"
    ep, ref, spec = _run(cfg, 8, split=True)
    assert LAST_KERNEL == "pointnet_split_persist", LAST_KERNEL


@pytest.mark.parametrize("N,B", [(1024, 8), (1000, 3), (333, 70), (128, 1), (4096, 2), (1024, 300)])
def test_split_persistent_kernel_is_bit_identical_to_the_tilewise_one(gpu_required, N, B):
    """pointnet_split_persist (persistent workgroups, two channel tiles per wave in the last layer, the hidden layer as the transposed product
    with packed stores; C3 = 256 / 512 / 1024 in the three stages of the shipped widths) accumulates every output in the order
    pointnet_split<64, 128> does: identical bits, on ragged tiles (N not a multiple of 128), fewer tiles than CUs and several passes of the
    grid; a second launch (the weight ring and the point prefetch start over) gives the same bits again."""
    cfg = alignnet3d.default_model_config()
    cfg["model"]["num_points"] = N
    spec, P32 = oracle_params(cfg)
    d = R.synth_pairs(B, N, seed=77, dtype=np.float32)
    out = {}
    for mode, kernel in (("tilewise", "pointnet_split<64,128>"), ("persist", "pointnet_split_persist")):
        eng = alignnet3d.Engine(cfg)
        eng.set_variables(P32)
        eng.set_option("infer_matmul_bf16x3", 1)
        eng.set_option("ab_split_tilewise", int(mode == "tilewise"))
        out[mode] = eng.forward(d["pcs1"], d["pcs2"])
        if mode == "persist":
            again = eng.forward(d["pcs1"], d["pcs2"])
            for k in again:
                np.testing.assert_array_equal(again[k], out[mode][k], err_msg=k)
        assert eng.last_backbone_kernel() == kernel
        eng.close()
    for k in out["persist"]:
        np.testing.assert_array_equal(out["persist"][k], out["tilewise"][k], err_msg=k)


@pytest.mark.parametrize("split", [False, True])
def test_forward_default_json_widths(gpu_required, split):
    # reference configs/default.json widths: 5-layer backbones, 36 bins (the split-bf16 option does not cover 5-layer
    # backbones or a 128-wide first layer: every stage must fall back to the exact kernels)
    cfg = alignnet3d.default_model_config()
    o = cfg["model"]["options"]
    o["s1transformer"] = [[128, 128, 256], [[512, 256], 0.7]]
    o["s2transformer"] = [[64, 64, 64, 128, 1024], [[512, 256], 0.7]]
    o["embedding"] = [64, 64, 64, 128, 1024]
    cfg["model"]["angles"]["num_bins"] = 36
    cfg["model"]["num_points"] = 512
    ep, ref, spec = _run(cfg, 4, split=split)
    worst, unstable = compare_forward(ep, ref, spec.num_bins)
    print("worst abs err", worst, "unstable pairs", unstable)


def test_get_set_roundtrip_and_errors(gpu_required):
    eng = alignnet3d.Engine(small_cfg())
    names = eng.variables()
    assert len(names) == len(R.param_names(R.NetSpec.from_cfg(small_cfg())))
    name, shp, _ = names[0]
    v = np.arange(shp[0] * shp[1], dtype=np.float32).reshape(shp) / 7
    eng.set_variable(name, v)
    np.testing.assert_array_equal(eng.get_variable(name), v)
    with pytest.raises(alignnet3d.EngineError, match="unknown variable"):
        eng.set_variable("nope/weights", v)
    with pytest.raises(alignnet3d.EngineError, match="elements"):
        eng.set_variable(name, v.ravel()[:-1])
    with pytest.raises(ValueError):
        eng.forward(np.zeros((2, 7, 3), np.float32), np.zeros((2, 7, 3), np.float32))
    eng.close()


def test_batch_independence_and_determinism(gpu_required):
    cfg = small_cfg(N=128)
    spec, P32 = oracle_params(cfg)
    eng = alignnet3d.Engine(cfg)
    eng.set_variables(P32)
    d = R.synth_pairs(9, 128, dtype=np.float32)
    full = eng.forward(d["pcs1"], d["pcs2"])
    again = eng.forward(d["pcs1"], d["pcs2"])
    part = eng.forward(d["pcs1"][2:5], d["pcs2"][2:5])
    for k in full:
        np.testing.assert_array_equal(full[k], again[k])          # bit-reproducible
        np.testing.assert_array_equal(full[k][2:5], part[k])      # eval-mode pairs are independent
    eng.close()


@pytest.mark.parametrize("N,B", [(64, 4), (128, 6), (200, 3)])
def test_forward_dgcnn(gpu_required, N, B, split=False):
    """DGCNN branch (reference models/tp8.py:30-46): static kNN graph (k = 20, self included), edge convs,
    max over neighbours, point conv, max over points.  The oracle rebuilds the graph per stage in fp64; the
    engine builds it once in the mean-centred frame in fp32 (the frames differ by a rigid motion).  A different
    neighbour at a near-tie changes an edge feature, so mismatching pairs are counted rather than tolerated."""
    cfg = small_cfg(N=N, backbone="dgcnn")
    ep, ref, spec = _run(cfg, B, split=split)
    bad = 0
    for b in range(B):
        ok = all(np.allclose(ep[k][b], ref[k][b], rtol=2e-4, atol=2e-4) for k in
                 ("pred_s1_pc1centers", "pred_s1_pc2centers", "pred_s2_pc1centers", "pred_s2_pc2centers", "pred_pc1angle_logits", "pred_pc2angle_logits"))
        bad += not ok
    print("dgcnn pairs outside 2e-4:", bad, "of", B)
    assert bad == 0


@pytest.mark.parametrize("N", [1500, 4096])
def test_forward_dgcnn_large_clouds(gpu_required, N):
    """The kNN kernel is compiled for 16 / 32 / 64 candidate slots per lane (N <= 1024 / 2048 / 4096): the small-N cases above
    only reach the first.  Same criterion; a neighbour swapped at a near-tie of the fp32 distances (the oracle's are fp64)
    would show up as a mismatching pair."""
    test_forward_dgcnn(gpu_required, N, 2)


@pytest.mark.parametrize("N,B", [(64, 3), (200, 4)])
def test_forward_dgcnn_split_bf16(gpu_required, N, B):
    """The DGCNN branch with the split-bf16 kernels (dgcnn_split): same criterion as the exact-fp32 branch."""
    test_forward_dgcnn(gpu_required, N, B, split=True)


DG_KEYS = ("pred_s1_pc1centers", "pred_s1_pc2centers", "pred_s2_pc1centers", "pred_s2_pc2centers", "pred_pc1angle_logits", "pred_pc2angle_logits")


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("N,B,full", [(1024, 4, True), (4096, 2, True), (4096, 2, False), (200, 5, False), (1500, 2, False)])
def test_forward_dgcnn_shipped_widths(gpu_required, N, B, full, split):
    """The DGCNN kernels bench.py times (BASELINE.json configs[4]): edge widths 64 / 128 dispatch to instantiations with the widths
    compiled in -- dgcnn_fused<68, 132> (two workgroups per CU, bounded to 128 VGPRs) and dgcnn_split<64, 128> -- which are different
    code from the generic kernels the small-width tests reach.  full = the SynthCars widths (C3 = 256 / 512 / 1024) at the bench's
    cloud sizes; otherwise narrow last layers so that ragged N (partial tiles, all three kNN slot counts) stays cheap for the
    oracle.  Same criterion as test_forward_dgcnn, plus the stage-3 outputs on decode-stable pairs, plus the read-back of which
    instantiation ran."""
    if full:
        cfg = alignnet3d.default_model_config()
        cfg["model"]["num_points"] = N
        cfg["model"]["backbone"] = "dgcnn"
    else:
        cfg = small_cfg(N=N, backbone="dgcnn", s1=(64, 128, 96), s2=(64, 128, 128), emb=(64, 128, 160))
    ep, ref, spec = _run(cfg, B, split=split)
    assert LAST_KERNEL == ("dgcnn_split<64,128>" if split else "dgcnn_fused<64,128>"), LAST_KERNEL
    bad = sum(not all(np.allclose(ep[k][b], ref[k][b], rtol=2e-4, atol=2e-4) for k in DG_KEYS) for b in range(B))
    worst, unstable = compare_forward(ep, ref, spec.num_bins, atol=2e-4, rtol=2e-4)
    print("dgcnn shipped widths N=%d B=%d split=%s: pairs outside 2e-4: %d, worst abs err %.2e, unstable %d" % (N, B, split, bad, max(worst.values()), unstable))
    assert bad == 0 and unstable <= 1


@pytest.mark.parametrize("split", [False, True])
def test_forward_whole_cloud_workgroups_with_partial_tile(gpu_required, split):
    """B = 256 makes a workgroup walk all tiles of its cloud (pooled max in registers, published once); N = 200 makes the
    last of its two tiles partial.  Same bar as every other forward test."""
    cfg = small_cfg(N=200)
    ep, ref, spec = _run(cfg, 256, split=split)
    worst, unstable = compare_forward(ep, ref, spec.num_bins)
    print("worst abs err", max(worst.values()), "unstable pairs", unstable)
    assert unstable <= 64


def _knn_graph(cfg, pcs1, pcs2):
    spec, P32 = oracle_params(cfg)
    eng = alignnet3d.Engine(cfg)
    eng.set_variables(P32)
    eng.forward(pcs1, pcs2)
    g = eng.debug_knn_graph(len(pcs1))
    eng.close()
    return g


@pytest.mark.parametrize("N,B", [(20, 2), (64, 3), (200, 3), (1024, 2), (1500, 2), (2048, 1), (3000, 1), (4096, 2)])
def test_knn_graph_against_oracle(gpu_required, N, B):
    """The neighbour graph itself (index work), read back through alignnet_debug_knn_graph, against the oracle's knn_indices
    (utils/tf_util_dgcnn.py:638-671) evaluated in fp64 on the mean-centred clouds.  The kernel's distances are fp32, so a query's
    list may differ from the fp64 one where two candidates are closer than fp32 resolves; the test therefore demands, for EVERY
    query: 20 distinct in-range indices, nearest first and nothing nearer left out -- both up to the fp32 rounding of the distance
    formula -- and counts the lists that are not identical to the oracle's (a handful per cloud at most).  N covers all three
    compiled slot counts (16 / 32 / 64 per lane), partial pair slots and a cloud smaller than one wave."""
    k = 20
    d = R.synth_pairs(B, N, seed=77 + N, dtype=np.float32)
    g = _knn_graph(small_cfg(N=N, backbone="dgcnn"), d["pcs1"], d["pcs2"])
    assert g.shape == (2, B, N, k) and g.min() >= 0 and g.max() < N
    differing = 0
    for t, pcs in enumerate((d["pcs1"], d["pcs2"])):
        x = pcs.astype(np.float64)
        x = x - x.mean(axis=1, keepdims=True)
        want = R.knn_indices(x, k)
        sq = (x * x).sum(-1)
        for b in range(B):
            dist = sq[b][:, None] - 2.0 * x[b] @ x[b].T + sq[b][None, :]
            got = g[t, b]
            dg = np.take_along_axis(dist, got, axis=1)
            eps = 64 * np.finfo(np.float32).eps * (sq[b][:, None] + sq[b].max())     # cancellation in |x|^2 - 2 x.y + |y|^2
            assert all(len(set(r)) == k for r in got), "duplicate neighbour"
            assert (np.diff(dg, axis=1) >= -eps).all(), "not nearest-first"
            kth = np.sort(dist, axis=1)[:, k - 1:k]
            assert (dg <= kth + eps).all(), "a listed neighbour is farther than the k-th nearest"
            differing += int((got != want[b]).any(axis=1).sum())
    print("kNN N=%d: %d of %d lists differ from the fp64 oracle's" % (N, differing, 2 * B * N))
    assert differing <= max(2, (2 * B * N) // 200)


@pytest.mark.parametrize("N", [64, 1536, 4096])
def test_knn_graph_ties_bit_exact(gpu_required, N):
    """Integer lattice points with an exactly representable, exactly zero mean: every distance is a small integer, exact in fp32, so
    the graph must equal the oracle's index for index -- including the order within the many exact ties, where tf.nn.top_k (and the
    oracle's stable sort) put the lower point index first."""
    rng = np.random.default_rng(5 + N)
    half = rng.integers(-6, 7, size=(2, N // 2, 3)).astype(np.float32)
    pcs = np.concatenate([half, -half], axis=1)                       # mean exactly 0 in any summation order
    pcs = pcs[:, rng.permutation(N)]
    g = _knn_graph(small_cfg(N=N, backbone="dgcnn"), pcs[:1], pcs[1:])
    want = R.knn_indices(pcs.astype(np.float64), 20)
    np.testing.assert_array_equal(g[0, 0], want[0])
    np.testing.assert_array_equal(g[1, 0], want[1])


def test_pipelined_host_path_equals_blocking_forward(gpu_required):
    """alignnet_forward_submit / _wait (pinned staging, copy-in of batch i + 1 under the forward of batch i, two batches in flight) give
    the blocking alignnet_forward's outputs bit for bit, in order, for batches of different sizes; a third submit without a wait and
    a wait with nothing in flight fail with a message."""
    cfg = small_cfg(N=256)
    spec, P32 = oracle_params(cfg)
    eng = alignnet3d.Engine(cfg)
    eng.set_variables(P32)
    batches = [R.synth_pairs(b, 256, seed=40 + i, dtype=np.float32) for i, b in enumerate((5, 9, 3, 9, 1, 7))]
    want = [eng.forward(d["pcs1"], d["pcs2"]) for d in batches]
    got = list(eng.forward_stream((d["pcs1"], d["pcs2"]) for d in batches))
    assert len(got) == len(want)
    for g, w_ in zip(got, want):
        for k in w_:
            np.testing.assert_array_equal(g[k], w_[k])
    with pytest.raises(alignnet3d.EngineError):
        eng.forward_wait()
    eng.forward_submit(batches[0]["pcs1"], batches[0]["pcs2"])
    eng.forward_submit(batches[1]["pcs1"], batches[1]["pcs2"])
    with pytest.raises(alignnet3d.EngineError):
        eng.forward_submit(batches[2]["pcs1"], batches[2]["pcs2"])
    a, b = eng.forward_wait(), eng.forward_wait()
    np.testing.assert_array_equal(a["pred_translations"], want[0]["pred_translations"])
    np.testing.assert_array_equal(b["pred_translations"], want[1]["pred_translations"])
    eng.close()
