"""CPU: both oracles against fixtures made by EXECUTING the reference's graph-building code (models/tp8.py get_model / get_loss,
utils/tf_util.py, utils/tf_util_dgcnn.py -- unmodified, build container only) on a NumPy stand-in for TensorFlow
(tests/golden/tf_standin.py, generator tests/golden/make_graph_golden.py).

The stand-in is a library stand-in written for this repository: parity stays FORMALLY UNPINNED.  What these tests
add is that the wiring -- variable names and sharing under `reuse=tf.AUTO_REUSE`, layer order, the [B] / [B,1] broadcasts of the
loss, the whole-batch tf.cond, EMA updates -- comes from running the reference's text, not from reading it."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from alignnet3d import tf_bundle as tb
from oracle import alignnet_ref as R
from oracle import alignnet_torch as T

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "graph_vectors.npz"))
META = json.load(open(os.path.join(HERE, "golden", "graph_vectors.json")))
LABELS = ("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles")
US = ("s1_0", "s2_0", "s1_1", "s2_1", "rem")   # graph-construction order of the five dropout layers
EP = ("pred_translations", "pred_remaining_angle_logits", "pred_s1_pc1centers", "pred_s1_pc2centers", "pred_s2_pc1centers",
      "pred_s2_pc2centers", "pred_pc1angle_logits", "pred_pc2angle_logits")
CASES = sorted(META)


def case_cfg(case):
    m = META[case]
    w = m["widths"]
    return {"data": {"num_channels": 3, "ntrain": 1000},
            "model": {"backbone": m["backbone"], "num_points": m["N"],
                      "options": {"angle_factor": m["angle_factor"], "early_stage_factor": m["early_stage_factor"],
                                  "s1transformer": [w["s1"], [w["fc"], 0.7]], "s2transformer": [w["s2"], [w["fc"], 0.7]],
                                  "embedding": w["emb"], "remaining_transform_prediction": [w["fc"], 0.7]},
                      "angles": {"num_bins": m["num_bins"], "accept_inverted_angle": m["accept_inverted_angle"]}},
            "training": {"batch_size": m["B"], "learning_rate": 0.005, "optimizer": {"optimizer": "adam"},
                         "lr_extension": {"mode": "decay", "per": "epoch", "step": 30, "rate": 0.5},
                         # bn_decay at step 0 = min(clip, 1 - init) = the fixture's 0.75 (train.py:159-174)
                         "bn_extension": {"mode": "decay", "per": "epoch", "step": 30, "rate": 0.5, "init": 1.0 - m["bn_decay"], "clip": 0.99}},
            "gpu_index": 0}


def tf_name(oracle_name):
    return tb.tf_shadow_name(oracle_name) or oracle_name


def oracle_params(case, dtype=np.float64):
    """The fixture's variables under the oracle's names: conv kernels HWIO -> [Cin, Cout] (the [1,3,1,C] first kernel -> [3, C])."""
    spec = R.NetSpec.from_cfg(case_cfg(case))
    P = {}
    for name, shp in R.param_names(spec):
        a = G["%s/f32/var/%s" % (case, tf_name(name))]
        P[name] = a.reshape(shp).astype(dtype)
    return spec, P


def inputs(case, dtype=np.float64):
    d = {k: G["%s/f32/in/%s" % (case, k)].astype(dtype) for k in ("pcs1", "pcs2") + LABELS}
    u = {k: G["%s/f32/dropout_u/%d" % (case, i)].astype(dtype) for i, k in enumerate(US)}
    return d, u


@pytest.mark.parametrize("case", CASES)
def test_variable_names_and_sharing(case):
    """The variables the reference's code creates when executed, against the name list the oracles / the engine / the checkpoint
    reader assume: shared weights + biases under `siamese/`, one beta / gamma / EMA pair per tower (`siamese/`,
    `siamese_1/` name scopes), EMA slots named by the variable scope + the moments op's name-scope path, pair head at top level."""
    spec = R.NetSpec.from_cfg(case_cfg(case))
    created = META[case]["variables"]
    names = [v["name"] for v in created]
    assert len(names) == len(set(names))
    expected = {tf_name(n): shp for n, shp in R.param_names(spec)}
    assert set(names) == set(expected), (sorted(set(names) - set(expected))[:4], sorted(set(expected) - set(names))[:4])
    for v in created:
        assert int(np.prod(v["shape"])) == int(np.prod(expected[v["name"]])), v
        assert v["trainable"] == (not v["name"].endswith("ExponentialMovingAverage")), v
        if v["name"].endswith("conv1/weights"):
            assert v["shape"][:2] == ([1, 3] if spec.backbone == "pointnet" else [1, 1]) and v["shape"][2] == (1 if spec.backbone == "pointnet" else 6)
    # sharing: no weights under siamese_1/, BN variables under both
    assert not [n for n in names if n.startswith("siamese_1/") and n.endswith(("weights", "biases"))]
    assert len([n for n in names if n.startswith("siamese_1/") and n.endswith("/bn/gamma")]) == \
        len([n for n in names if n.startswith("siamese/") and n.endswith("/bn/gamma")])
    # creation order = graph order: tower 0 creates every shared variable, then tower 1 only its BN sets, then the pair head
    first_t1 = min(i for i, n in enumerate(names) if "siamese_1/" in n)
    assert all(("siamese_1/" in n) or n.startswith("fc") for n in names[first_t1:])
    assert META[case]["summary_tags"] == list(R.get_loss.__doc__ and [
        "losses/translation", "losses/angle", "losses_stages/stage1_pc1_transl_loss", "losses_stages/stage1_pc2_transl_loss",
        "losses_stages/stage2_pc1_transl_loss", "losses_stages/stage2_pc2_transl_loss", "losses_stages/stage3_transl_loss",
        "losses_stages/stage2_pc1_angle_loss", "losses_stages/stage2_pc1_angle_class_loss", "losses_stages/stage2_pc1_angle_residual_loss",
        "losses_stages/stage2_pc2_angle_loss", "losses_stages/stage2_pc2_angle_class_loss", "losses_stages/stage2_pc2_angle_residual_loss",
        "losses_stages/stage3_angle_loss", "losses_stages/stage3_angle_class_loss", "losses_stages/stage3_angle_residual_loss"])


def _check(case, prec, mode, ep, loss, summ, ema, tol):
    pre = "%s/%s/%s" % (case, prec, mode)
    worst = 0.0
    for k in EP:
        ref = G["%s/ep/%s" % (pre, k)].astype(np.float64)
        err = float(np.abs(np.asarray(ep[k], np.float64) - ref).max())
        worst = max(worst, err)
        assert err <= tol * max(1.0, float(np.abs(ref).max())), (case, prec, mode, k, err)
    ref_loss = float(G[pre + "/loss"])
    assert abs(float(loss) - ref_loss) <= tol * max(1.0, abs(ref_loss)), (case, prec, mode, float(loss), ref_loss)
    ref_s = G[pre + "/summaries"].astype(np.float64)
    for tag, r in zip(META[case]["summary_tags"], ref_s):
        assert abs(float(summ[tag]) - r) <= tol * max(1.0, abs(r)), (case, prec, mode, tag, float(summ[tag]), r)
    if mode == "train":
        n = 0
        for name, v in ema.items():
            ref = G["%s/ema_after/%s" % (pre, tf_name(name))].astype(np.float64)
            assert np.abs(np.asarray(v, np.float64).ravel() - ref.ravel()).max() <= tol * max(1.0, float(np.abs(ref).max())), (case, name)
            n += 1
        assert n == sum(1 for v in META[case]["variables"] if not v["trainable"])
    return worst


@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("case", CASES)
def test_numpy_oracle_matches_executed_reference_graph(case, mode):
    """oracle/alignnet_ref.py in fp64 against the wide (fp64) execution of the reference graph: 1e-11 (summation order only -- the
    first version of this test found four float32 constants, np.pi / nb, `+ np.pi`, keep_prob and the BN epsilon, that the oracles
    held in double: 1e-7 apart in training mode),
    and in fp32 against the fp32 execution: 2e-4."""
    for prec, dt, tol in (("f64", np.float64, 1e-11), ("f32", np.float32, 2e-4)):
        spec, P = oracle_params(case, dt)
        d, u = inputs(case, dt)
        train = mode == "train"
        ep, upd, _ = R.get_model(P, spec, d["pcs1"], d["pcs2"], is_training=train, bn_decay=dt(META[case]["bn_decay"]), dropout_u=u if train else None)
        loss, summ = R.get_loss(spec, ep, *[d[k] for k in LABELS])
        ema = {}
        if train:
            ema = {k: v for k, v in upd.items()}
        worst = _check(case, prec, mode, ep, loss, summ, ema, tol)
        print(case, mode, prec, "numpy oracle worst end-point error", worst)


@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("case", CASES)
def test_torch_oracle_matches_executed_reference_graph(case, mode):
    """oracle/alignnet_torch.py (the autograd reference of the HIP backward) in fp64 against the same fixtures."""
    spec, P = oracle_params(case, np.float64)
    d, u = inputs(case, np.float64)
    tm = T.TorchTp8(spec, T.to_torch(P))
    td = {k: torch.tensor(v) for k, v in d.items()}
    tu = {k: torch.tensor(v) for k, v in u.items()}
    train = mode == "train"
    with torch.no_grad():
        ep = tm.forward(td["pcs1"], td["pcs2"], train, META[case]["bn_decay"], tu if train else None)
        loss = tm.loss(ep, *[td[k] for k in LABELS])
    ep = {k: v.numpy() for k, v in ep.items()}
    pre = "%s/f64/%s" % (case, mode)
    for k in EP:
        ref = G["%s/ep/%s" % (pre, k)]
        assert np.abs(ep[k] - ref).max() <= 1e-6 * max(1.0, float(np.abs(ref).max())), (case, mode, k)
    assert abs(float(loss) - float(G[pre + "/loss"])) <= 1e-6
    if train:
        for name, v in tm.ema_updates.items():
            ref = G["%s/ema_after/%s" % (pre, tf_name(name))]
            assert np.abs(v.numpy().ravel() - ref.ravel()).max() <= 1e-10 * max(1.0, float(np.abs(ref).max())), name


def test_fixture_covers_both_tf_cond_branches_and_broadcasts():
    """The fixtures must exercise what they claim: accept_inverted on and off, a stage-3 target that is a genuine [B,B] matrix
    (its rows differ), dropout masks that drop something, towers with different BN sets."""
    assert {META[c]["accept_inverted_angle"] for c in CASES} == {True, False}
    assert {META[c]["backbone"] for c in CASES} == {"pointnet", "dgcnn"}
    for c in CASES:
        u = G["%s/f32/dropout_u/0" % c]
        assert 0.1 < float((np.floor(0.7 + u) == 0).mean()) < 0.5
        a = G["%s/f32/train/ep/pred_translations" % c]
        b = G["%s/f32/eval/ep/pred_translations" % c]
        assert not np.allclose(a, b)      # batch statistics + dropout vs shadows: the two modes differ
        g0 = G["%s/f32/var/siamese/transformer1/embedding/conv1/bn/gamma" % c]
        g1 = G["%s/f32/var/siamese_1/transformer1/embedding/conv1/bn/gamma" % c]
        assert not np.array_equal(g0, g1)


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="the reference only exists in the build container")
def test_fixtures_regenerate_bit_identically(tmp_path):
    """Re-run the generator against /root/reference (build container only) and compare with the committed fixtures."""
    gen = os.path.join(HERE, "golden", "make_graph_golden.py")
    code = ("import sys, runpy; sys.argv=['x']; import numpy as np; import os\n"
            "g = runpy.run_path(%r)\n" % gen)
    env = dict(os.environ, ALIGNNET_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, gen], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    new = np.load(os.path.join(str(tmp_path), "graph_vectors.npz"))
    assert sorted(new.files) == sorted(G.files)
    for k in G.files:
        np.testing.assert_array_equal(new[k], G[k], err_msg=k)
    assert json.load(open(os.path.join(str(tmp_path), "graph_vectors.json"))) == META
