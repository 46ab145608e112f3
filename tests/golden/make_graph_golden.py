#!/usr/bin/env python3
"""Execute the REFERENCE's graph-building code (models/tp8.py get_model + get_loss with utils/tf_util.py and
utils/tf_util_dgcnn.py, imported unmodified from /root/reference -- build container only) against the NumPy stand-in for
TensorFlow in tests/golden/tf_standin.py, on small shapes, and write what it computed as fixtures:

    tests/golden/graph_vectors.npz / graph_vectors.json
      per case: the variable list the code created (TF names, shapes, trainable flags, creation order), every variable's
      value, the eight fed arrays, and for is_training = False and True: the eight end_points, the loss, the 16 summaries of
      models/tp8.py:336-353, the dropout uniforms in graph-construction order, and the EMA shadows after the training run.

READ tests/golden/tf_standin.py's header: this is a stand-in for a library, parity stays formally unpinned.  The fixtures
machine-check the reference's WIRING (scopes and sharing, layer order, broadcasts, tf.cond choices) against both oracles
(tests/test_graph_golden.py) and the HIP engine (tests/test_graph_golden_gpu.py).  Inputs are synthetic."""
import importlib
import json
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)


def _synth_pairs():
    """alignnet3d/synth.py loaded by path: the package directory must NOT be on sys.path here, it holds this repository's own
    `models`, `config` and `provider`, which would shadow the reference's."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_synth", os.path.join(ROOT, "alignnet-3d_amd", "alignnet3d", "synth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.synth_pairs

CASES = {
    # name: (backbone, B, N, model options, num_bins, accept_inverted_angle)
    "pointnet": ("pointnet", 6, 64, dict(s1=[32, 64, 96], s2=[32, 64, 128], emb=[32, 64, 160], fc=[64, 32]), 12, True),
    "pointnet_noinv_deep": ("pointnet", 5, 48, dict(s1=[16, 16, 32, 48], s2=[16, 32, 64], emb=[16, 16, 16, 32, 64], fc=[32, 16]), 10, False),
    "dgcnn": ("dgcnn", 4, 40, dict(s1=[32, 64, 96], s2=[32, 64, 128], emb=[64, 128, 160], fc=[64, 32]), 12, True),
}
LABELS = ("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles")


def build_and_run(case, wide, tmp):
    import tf_standin as ts
    synth_pairs = _synth_pairs()
    backbone, B, N, w, nb, inv = CASES[case]
    ts.set_float(np.float64 if wide else np.float32)
    ts.reset_default_graph()
    ts.seed_initializers(1000 + sorted(CASES).index(case))
    import config
    cfg = config.configGlobal
    user = {"data": {"basepath": os.path.join(tmp, "TinySet")}, "logging": {"basedir": os.path.join(tmp, "logs")},
            "model": {"backbone": backbone, "num_points": N, "angles": {"num_bins": nb, "accept_inverted_angle": inv},
                      "options": {"angle_factor": 1.5, "early_stage_factor": 0.25,
                                  "s1transformer": [w["s1"], [w["fc"], 0.7]], "s2transformer": [w["s2"], [w["fc"], 0.7]],
                                  "embedding": w["emb"], "remaining_transform_prediction": [w["fc"], 0.7]}},
            "training": {"batch_size": B}}
    path = os.path.join(tmp, "Case_%s.json" % case)
    json.dump(user, open(path, "w"))
    config.load_config(path)
    tp8 = importlib.import_module("models.tp8")
    assert os.path.realpath(tp8.__file__).startswith(REF) and os.path.realpath(config.__file__).startswith(REF), "not the reference's modules"
    tf = sys.modules["tensorflow"]
    ph = tp8.placeholder_inputs(B, N)                                   # train.py:190
    is_training = tf.placeholder(tf.bool, shape=())                       # train.py:191
    bn_decay = tf.placeholder(tf.float32, shape=())                       # train.py:196 hands get_model a scalar tensor
    end_points = tp8.get_model(ph[0], ph[1], is_training, bn_decay=bn_decay)   # train.py:200
    loss = tp8.get_loss(*ph, end_points)                                  # train.py:201
    g = ts.get_default_graph()
    sess = ts.Session()
    sess.init_variables()
    # non-trivial BN variables / shadows / biases, as a trained checkpoint would hold (each tower's differ)
    rng = np.random.default_rng(7)
    for v in g.variables:
        n, shp = v.var_name, np.shape(v.initial)
        if n.endswith("/beta"):
            val = rng.normal(0, 0.1, shp)
        elif n.endswith("/gamma"):
            val = rng.uniform(0.5, 1.5, shp)
        elif n.endswith("/biases"):
            val = rng.normal(0, 0.05, shp)
        elif n.endswith("Squeeze/ExponentialMovingAverage"):
            val = rng.normal(0, 0.2, shp)
        elif n.endswith("Squeeze_1/ExponentialMovingAverage"):
            val = rng.uniform(0.5, 2.0, shp)
        else:
            continue
        sess.values[v] = val.astype(np.float32).astype(ts.FLOAT)
    out = {}
    var_list = [{"name": v.var_name, "shape": list(np.shape(v.initial)), "trainable": bool(v.trainable)} for v in g.variables]
    for v in g.variables:
        out["var/" + v.var_name] = np.asarray(sess.values[v], np.float32)
    d = synth_pairs(B, N, seed=4242, dtype=np.float32)
    feeds = dict(zip(ph, [d["pcs1"], d["pcs2"]] + [d[k] for k in LABELS]))
    for k in ("pcs1", "pcs2") + LABELS:
        out["in/" + k] = d[k]
    tags = [t for t, _ in g.summaries]
    fetch = {"ep": end_points, "loss": loss, "summ": [t for _, t in g.summaries]}
    drops = g.collections.get("_dropout_nodes", [])
    urng = np.random.default_rng(99)
    uniforms = [urng.uniform(size=np.shape(n.build_value)).astype(np.float32) for n in drops]
    for i, u in enumerate(uniforms):
        out["dropout_u/%d" % i] = u
    decay = 0.75
    for mode in ("eval", "train"):
        feeds[is_training] = np.asarray(mode == "train")
        feeds[bn_decay] = np.asarray(decay, np.float32)
        sess.dropout_uniforms = dict(zip(drops, uniforms))
        before = {v: np.array(sess.values[v]) for v in g.variables}
        r = sess.run(fetch, feeds)
        for k, v in r["ep"].items():
            out["%s/ep/%s" % (mode, k)] = np.asarray(v)
        out[mode + "/loss"] = np.asarray(r["loss"])
        out[mode + "/summaries"] = np.asarray([np.asarray(x) for x in r["summ"]])
        changed = [v for v in g.variables if not np.array_equal(before[v], sess.values[v])]
        if mode == "eval":
            assert not changed and not sess.drawn, "eval mode must not touch variables or draw dropout masks"
        else:
            assert len(sess.drawn) == len(drops) == 5
            assert all(c.var_name.endswith("ExponentialMovingAverage") for c in changed)
            for v in g.variables:
                if v.var_name.endswith("ExponentialMovingAverage"):
                    out["train/ema_after/" + v.var_name] = np.asarray(sess.values[v])
    meta = {"backbone": backbone, "B": B, "N": N, "num_bins": nb, "accept_inverted_angle": inv, "widths": w, "bn_decay": decay,
            "angle_factor": 1.5, "early_stage_factor": 0.25, "variables": var_list, "summary_tags": tags,
            "dropout_order": ["s1 tower 0", "s2 tower 0", "s1 tower 1", "s2 tower 1", "pair head"]}
    return out, meta


def main():
    import tf_standin as ts
    ts.install()
    sys.path.insert(0, os.path.join(REF, "tp_utils"))
    sys.path.insert(0, REF)
    for name in ("quaternion", "pyntcloud", "open3d", "pythreejs", "trimesh", "IPython", "IPython.display", "ipywidgets"):
        from unittest.mock import MagicMock
        sys.modules[name] = MagicMock()
    import make_golden
    arrays, metas = {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        make_golden.make_dataset(os.path.join(tmp, "TinySet"), np.random.default_rng(1))
        import provider  # noqa: F401  (must precede config: circular import in the reference)
        for case in CASES:
            for wide in (False, True):
                out, meta = build_and_run(case, wide, tmp)
                tag = "%s/%s" % (case, "f64" if wide else "f32")
                for k, v in out.items():
                    if wide and (k.startswith(("var/", "in/", "dropout_u/"))):
                        continue   # identical to the f32 run's (float32-representable by construction)
                    arrays[tag + "/" + k] = v
                metas[case] = meta
    dst = os.environ.get("ALIGNNET_GOLDEN_OUT", HERE)   # tests/test_graph_golden.py regenerates into a scratch directory
    np.savez_compressed(os.path.join(dst, "graph_vectors.npz"), **arrays)
    json.dump(metas, open(os.path.join(dst, "graph_vectors.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(arrays), "arrays for", list(metas))


if __name__ == "__main__":
    main()
