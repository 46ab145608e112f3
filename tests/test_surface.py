"""CPU: the Python surface of the drop-in (config / provider / decode / evaluation helpers) against golden
vectors produced by the REFERENCE's own functions (tests/golden/make_golden.py, run in the build container)."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "reference_vectors.npz"))
J = json.load(open(os.path.join(HERE, "golden", "reference_vectors.json")))
PKG = os.path.join(os.path.dirname(HERE), "alignnet-3d_amd")


@pytest.fixture()
def tiny(tmp_path):
    """Re-materialise the fixture dataset + config, import the drop-in modules fresh."""
    root = tmp_path / "TinySet"
    for sub in ("meta", "pointcloud1", "pointcloud2", "split"):
        (root / sub).mkdir(parents=True)
    for i, meta in J["ds_meta"].items():
        json.dump(meta, open(root / "meta" / ("%08d.json" % int(i)), "w"))
        np.save(root / "pointcloud1" / ("%08d.npy" % int(i)), G["ds_pc1_%s" % i])
        np.save(root / "pointcloud2" / ("%08d.npy" % int(i)), G["ds_pc2_%s" % i])
    (root / "split" / "train.txt").write_text("\n".join(map(str, J["ds_train"])) + "\n")
    (root / "split" / "val.txt").write_text("\n".join(map(str, J["ds_val"])) + "\n")
    user = json.loads(json.dumps(J["config_user"]).replace(J["config_user"]["data"]["basepath"], str(root)))
    user["logging"]["basedir"] = str(tmp_path / "logs")
    cfg_path = tmp_path / "TinyRun.json"
    json.dump(user, open(cfg_path, "w"))
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    for m in ("config", "provider", "models.tp8", "models", "evaluation"):
        sys.modules.pop(m, None)
    config = importlib.import_module("config")
    config.load_config(str(cfg_path))
    return dict(tmp=str(tmp_path), config=config, provider=importlib.import_module("provider"),
                tp8=importlib.import_module("models.tp8"), evaluation=importlib.import_module("evaluation"))


def test_config_merge_matches_reference(tiny):
    merged = json.loads(json.dumps(tiny["config"].configGlobal.to_dict()).replace(tiny["tmp"], J["tmp_token"]))
    assert merged == J["config_merged"]
    cfg = tiny["config"].configGlobal
    assert cfg.has("name") and cfg.evaluation.has("special") and not cfg.has("nope")
    out = os.path.join(tiny["tmp"], "saved.json")
    tiny["config"].save_config(out)
    assert json.load(open(out)) == tiny["config"].configGlobal.to_dict()


def test_provider_batches_match_reference(tiny):
    prov = tiny["provider"]
    assert prov.getDataFiles(tiny["config"].configGlobal.data.basepath + "/split/train.txt") == J["ds_train"]
    names = ("pcs1", "pcs2", "translations", "rel_angles", "pc1centers", "pc2centers", "pc1angles", "pc2angles")
    np.random.seed(1234)
    b = prov.load_batch([0, 1, 2, 3])
    for k, v in zip(names, b):
        assert v.dtype == np.float64
        np.testing.assert_array_equal(v, G["lb_" + k], err_msg=k)
    np.random.seed(99)
    b2 = prov.load_batch([6, 5], override_batch_size=3)
    for k, v in zip(names, b2):
        assert v.shape[0] == 3
        np.testing.assert_array_equal(v[:2], G["lb2_" + k], err_msg=k)
    b3 = prov.load_batch([4, 0], override_batch_size=2, dont_load_pointclouds=True)
    np.testing.assert_array_equal(b3[2], G["lb3_translations"])
    np.testing.assert_array_equal(b3[7], G["lb3_pc2angles"])
    np.random.seed(7)
    np.testing.assert_array_equal(prov.jitter_point_cloud(G["jit_in"].copy()), G["jit_out"])


def test_host_decode_matches_reference(tiny):
    tp8 = tiny["tp8"]
    np.testing.assert_array_equal(tp8.classLogits2angle(G["dec_logits"]), G["dec_angles"])
    got = np.array([tp8.class2angle(c, r) for c, r in ((0, 0.1), (5, 0.3), (9, 0.9), (7, -0.2))])
    np.testing.assert_array_equal(got, G["dec_class2angle"])
    # the oracle's restatement of the same function
    from oracle import alignnet_ref as R
    np.testing.assert_array_equal(R.class_logits_to_angle(G["dec_logits"], 10), G["dec_angles"])
    ph = tp8.placeholder_inputs(4, 32)
    assert [p.shape for p in ph] == [(4, 32, 3), (4, 32, 3), (4, 3), (4, 1), (4, 3), (4, 3), (4, 1), (4, 1)]


def test_evaluation_helpers_match_reference(tiny):
    ev = tiny["evaluation"]
    t, gt, a, ga = G["ev_t"], G["ev_gt_t"], G["ev_a"], G["ev_gt_a"]
    np.testing.assert_allclose([ev.eval_translation(x, y)[0] for x, y in zip(t, gt)], G["ev_transl_dist"], rtol=0, atol=0)
    np.testing.assert_array_equal([ev.eval_translation(x, y)[1] for x, y in zip(t, gt)], G["ev_transl_lvl"])
    for inv in (False, True):
        np.testing.assert_allclose([ev.eval_angle(x, y, inv)[0] for x, y in zip(a, ga)], G["ev_angle_dist_%d" % inv], rtol=1e-15)
        np.testing.assert_array_equal([ev.eval_angle(x, y, inv)[1] for x, y in zip(a, ga)], G["ev_angle_lvl_%d" % inv])
        np.testing.assert_array_equal([ev.eval_transform(x, y, p, q, inv) for x, y, p, q in zip(t, gt, a, ga)], G["ev_transform_%d" % inv])
    np.testing.assert_allclose(ev.translate_transform_to_new_center_of_rotation(G["ttc_in_t"], G["ttc_in_a"], G["ttc_in_c"], G["ttc_in_g"]),
                               G["ttc_out"], rtol=1e-13, atol=1e-13)
    np.testing.assert_array_equal(tiny["provider"].str_to_np(J["np_to_str"]), G["str_to_np"])


def test_evaluate_writes_reference_schema(tiny):
    ev, cfg = tiny["evaluation"], tiny["config"].configGlobal
    val = J["ds_val"]
    n = len(val)
    rng = np.random.default_rng(0)
    out_dir = os.path.join(tiny["tmp"], "evaldir")
    res = ev.evaluate(cfg, val, rng.normal(size=(n, 3)).astype(np.float32) * 0.05, rng.normal(size=(n, 1)).astype(np.float32) * 0.05,
                      np.zeros((n, 3), np.float32), np.zeros((n, 1), np.float32), np.zeros((n, 3), np.float32),
                      np.ones((n, 3), np.float32), eval_dir=out_dir, accept_inverted_angle=True, mean_time=0.5)
    j = json.load(open(os.path.join(out_dir, "eval_180.json")))
    for key in ("corr_levels", "corr_levels_translation", "mean_dist_translation", "mean_sq_dist_translation", "corr_levels_angles",
                "mean_dist_angle", "mean_sq_dist_angle", "num", "eval_5m", "eval_10m", "eval_15m", "eval_20m", "val", "test", "reg_eval",
                "mean_time"):
        assert key in j, key
    assert j["num"] == n and j["mean_time"] == 0.5 and res.val.num == n and res.test.num == 0


def test_cli_surface():
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    src = open(os.path.join(PKG, "train.py")).read()
    for flag in ("--config", "--refineICP", "--its", "--use_old_results", "--refineICPmethod", "--eval_epoch"):
        assert flag in src
    for fname in ("pred_translations", "pred_angles", "pred_s1_pc1centers", "pred_s1_pc2centers", "pred_s2_pc1centers",
                  "pred_s2_pc2centers", "pred_s2_pc1angles", "pred_s2_pc2angles"):
        assert fname in src


def test_packed_loader_is_bit_identical_to_reference_batches(tiny):
    """the packed cache must reproduce the reference loader's seeded batches exactly."""
    prov = tiny["provider"]
    ds = prov.use_packed_cache()
    assert len(ds) == len(J["ds_meta"])
    names = ("pcs1", "pcs2", "translations", "rel_angles", "pc1centers", "pc2centers", "pc1angles", "pc2angles")
    np.random.seed(1234)
    b = prov.load_batch([0, 1, 2, 3])
    for k, v in zip(names, b):
        assert v.dtype == np.float64
        np.testing.assert_array_equal(v, G["lb_" + k], err_msg=k)
    np.random.seed(99)
    b2 = prov.load_batch([6, 5], override_batch_size=3)
    for k, v in zip(names, b2):
        np.testing.assert_array_equal(v[:2], G["lb2_" + k], err_msg=k)
    b3 = prov.load_batch([4, 0], override_batch_size=2, dont_load_pointclouds=True)
    np.testing.assert_array_equal(b3[2], G["lb3_translations"])
    np.testing.assert_array_equal(b3[7], G["lb3_pc2angles"])
    prov._packed = None


def test_packed_cache_is_rebuilt_when_the_dataset_changes(tiny):
    """The cache carries a manifest (example count, newest modification time); a changed dataset must not silently reuse it,
    and a build never leaves a half-written table under its final name."""
    from alignnet3d import packed
    prov, cfg = tiny["provider"], tiny["config"].configGlobal
    base = cfg.data.basepath
    cache = os.path.join(base, "packed_cache")
    prov.use_packed_cache()
    assert packed.cache_is_current(base, cache)
    man = json.load(open(os.path.join(cache, "manifest.json")))
    assert man["n_examples"] == len(J["ds_meta"])
    assert not [f for f in os.listdir(cache) if ".tmp" in f]
    stamp = os.stat(os.path.join(cache, "ids.npy")).st_mtime_ns
    prov.use_packed_cache()                                  # current: not rebuilt
    assert os.stat(os.path.join(cache, "ids.npy")).st_mtime_ns == stamp
    # a new example appears in the dataset
    src = sorted(os.listdir(os.path.join(base, "meta")))[0]
    new = "%08d" % 9999
    import shutil
    for sub, ext in (("meta", ".json"), ("pointcloud1", ".npy"), ("pointcloud2", ".npy")):
        shutil.copy(os.path.join(base, sub, src[:-5] + ext), os.path.join(base, sub, new + ext))
    try:
        assert not packed.cache_is_current(base, cache)
        ds = prov.use_packed_cache()
        assert len(ds) == len(J["ds_meta"]) + 1 and packed.cache_is_current(base, cache)
    finally:
        for sub, ext in (("meta", ".json"), ("pointcloud1", ".npy"), ("pointcloud2", ".npy")):
            os.remove(os.path.join(base, sub, new + ext))
        prov._packed = None
    assert not packed.cache_is_current(base, cache)


# ---- evaluation.evaluate against one full run of the REFERENCE's evaluate (tests/golden/make_golden.py eval_fixtures) -----------------
EV = np.load(os.path.join(HERE, "golden", "eval_vectors.npz"))
EJ = json.load(open(os.path.join(HERE, "golden", "eval_vectors.json")))


def _close(a, b, path=""):
    if isinstance(b, dict):
        assert isinstance(a, dict) and sorted(a) == sorted(b), path
        for k in b:
            _close(a[k], b[k], path + "/" + k)
    elif isinstance(b, list):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _close(x, y, "%s[%d]" % (path, i))
    elif isinstance(b, float):
        assert abs(a - b) <= 1e-12 * max(1.0, abs(b)), (path, a, b)
    else:
        assert a == b, (path, a, b)


@pytest.mark.parametrize("kind", ["synth", "kitti"])
def test_evaluate_numbers_match_the_reference(tiny, tmp_path, kind):
    """eval.json / eval_180.json as the reference's evaluation.evaluate (evaluation.py:128-289) writes them for the same predictions:
    every accuracy level, mean / rms distance, range bucket (5 / 10 / 15 / 20 m), the val / test split (sample index >= 1000 for
    Synth* base paths, tracklet ids for KITTI_tracklets*), the > 10 km outlier skip, the per-transform details and the velocity
    track files (evaluation.py:82-110, 213-228)."""
    ev, cfg = tiny["evaluation"], tiny["config"].configGlobal
    root = os.path.join(str(tmp_path), EJ[kind + "_dirname"])
    os.makedirs(os.path.join(root, "meta"))
    for i, m in enumerate(EJ[kind + "_meta"]):
        json.dump(m, open(os.path.join(root, "meta", "%08d.json" % i), "w"))
    old = cfg.data.basepath
    vars(cfg.data)["basepath"] = root
    try:
        for inv in (0, 1):
            ed = os.path.join(str(tmp_path), "out_%s_%d" % (kind, inv))
            res, detail = ev.evaluate(cfg, [int(v) for v in EV[kind + "_val"]], EV[kind + "_pred_t"], EV[kind + "_pred_a"], EV[kind + "_gt_t"],
                                      EV[kind + "_gt_a"], EV[kind + "_pred_c"], EV[kind + "_gt_c"], eval_dir=ed, accept_inverted_angle=bool(inv),
                                      detailed_eval=True, mean_time=0.25)
            written = json.load(open(os.path.join(ed, "eval_180.json" if inv else "eval.json")))
            _close(written, EJ["%s_eval_%d" % (kind, inv)], "eval")
            np.testing.assert_array_equal(np.array([d[0] for d in detail]), EV["%s_detail_levels_%d" % (kind, inv)])
            np.testing.assert_allclose(np.array([[d[1], d[2]] for d in detail]), EV["%s_detail_dists_%d" % (kind, inv)], rtol=1e-12, atol=1e-12)
            want = EJ["%s_velocity_files_%d" % (kind, inv)]
            vdir = os.path.join(ed, "velocities")
            got = {f: [float(x) for x in open(os.path.join(vdir, f)).read().split()] for f in sorted(os.listdir(vdir))} if os.path.isdir(vdir) else {}
            assert sorted(got) == sorted(want)
            for f in want:
                np.testing.assert_allclose(got[f], want[f], rtol=1e-12, atol=1e-15, err_msg=f)
        assert (kind == "kitti") == bool(want)
    finally:
        vars(cfg.data)["basepath"] = old


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="the reference only exists in the build container")
def test_every_shipped_reference_config_marshals(tiny, monkeypatch):
    """Each configs/*.json of the reference (read in place, build container only) through the drop-in's config.py merge and the
    engine's config marshalling (make_c_config): num_points / bins / widths / keep probabilities / schedules arrive as written,
    and nothing that the engine does not build (loss variants, backbones) slips through silently."""
    import glob
    from alignnet3d.engine import make_c_config
    cfgmod = tiny["config"]
    monkeypatch.setattr(cfgmod, "_read_split", lambda path: list(range(1000)))   # the datasets are not here: ntrain / nval only
    seen = 0
    for path in sorted(glob.glob("/root/reference/configs/*.json")):
        user = json.load(open(path))
        cfgmod.reset_config()
        cfg = cfgmod.load_config(path)
        assert cfg.name == os.path.basename(path)[:-5] and cfg.data.ntrain == 1000
        c = make_c_config(cfg)
        m = cfg.to_dict()["model"]
        assert c.num_points == m["num_points"] and c.num_bins == m["angles"]["num_bins"]
        assert c.backbone == {"pointnet": 0, "dgcnn": 1}[m["backbone"]]
        assert list(c.s1_conv.w[:c.s1_conv.n]) == m["options"]["s1transformer"][0]
        assert list(c.s2_conv.w[:c.s2_conv.n]) == m["options"]["s2transformer"][0]
        assert list(c.emb_conv.w[:c.emb_conv.n]) == m["options"]["embedding"]
        assert list(c.rem_fc.w[:c.rem_fc.n]) == m["options"]["remaining_transform_prediction"][0]
        assert abs(c.s1_keep - m["options"]["s1transformer"][1][1]) < 1e-7
        assert c.batch_size == cfg.training.batch_size and abs(c.learning_rate - cfg.training.learning_rate) < 1e-9
        assert c.accept_inverted_angle == int(m["angles"]["accept_inverted_angle"])
        if "model" in user and "num_points" in user["model"]:
            assert c.num_points == user["model"]["num_points"]      # the user file wins over default.json
        seen += 1
    assert seen >= 8
    cfgmod.reset_config()


def test_own_bench_configs_marshal(tiny, monkeypatch):
    """alignnet-3d_amd/configs/*.json (this build's own files: the BASELINE.json workloads in the reference's schema) load through
    config.py and marshal into the C config; the non-reference keys they use (training.matmul_dtype, training.sync_bn, training.global_loss) survive the merge."""
    import glob
    from alignnet3d.engine import make_c_config
    cfgmod = tiny["config"]
    monkeypatch.setattr(cfgmod, "_read_split", lambda path: list(range(4096)))
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alignnet-3d_amd", "configs")
    paths = sorted(glob.glob(os.path.join(here, "*.json")))
    assert len(paths) >= 4
    for path in paths:
        cfgmod.reset_config()
        cfg = cfgmod.load_config(path)
        c = make_c_config(cfg)
        assert c.num_points in (1024, 4096) and c.num_bins == 50 and list(c.emb_conv.w[:3]) == [64, 128, 1024]
        if "dgcnn" in path:
            assert c.backbone == 1 and c.num_points == 4096
        if "bf16" in path:
            assert cfg.training.matmul_dtype == "bf16"
        if "b2048" in path:
            assert cfg.training.sync_bn is True and cfg.training.global_loss is True and c.batch_size == 2048
    cfgmod.reset_config()
