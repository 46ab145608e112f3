#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE implementation (run only in the build container, where
/root/reference exists; the GPU box never sees the reference).  What can be imported without TensorFlow
(SURVEY.md 8c): config.load_config/save_config, provider.load_batch/jitter_point_cloud/getDataFiles,
models.tp8.class2angle/classLogits2angle, evaluation.eval_translation/eval_angle/eval_transform,
pointcloud.np_to_str/str_to_np/get_mat_angle/translate_transform_to_new_center_of_rotation.

Outputs (committed): tests/golden/reference_vectors.npz + reference_vectors.json.
Inputs are synthetic (made here); only the OUTPUTS come from the reference code."""
import io
import json
import os
import sys
import tempfile
import types
from unittest.mock import MagicMock

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def stub_modules():
    for name in ("tensorflow", "tensorflow.python", "tensorflow.python.util", "tensorflow.python.util.nest", "quaternion", "pyntcloud",
                 "open3d", "pythreejs", "trimesh", "IPython", "IPython.display", "ipywidgets", "tf_util", "tf_util_dgcnn"):
        sys.modules[name] = MagicMock()
    # scipy's Rotation lost as_dcm/from_dcm; the reference calls them (tp_utils/pointcloud.py:287)
    import scipy.spatial.transform as sst

    class Rot:
        def __init__(self, r):
            self._r = r

        @staticmethod
        def from_rotvec(v):
            return Rot(sst.Rotation.from_rotvec(v))

        @staticmethod
        def from_dcm(m):
            return Rot(sst.Rotation.from_matrix(m))

        def as_dcm(self):
            return self._r.as_matrix()

        def as_euler(self, *a, **k):
            return self._r.as_euler(*a, **k)

    mod = types.ModuleType("scipy.spatial.transform")
    mod.Rotation = Rot
    sys.modules["scipy.spatial.transform"] = mod


def make_dataset(root, rng, n=7):
    """Tiny dataset in the reference's on-disk layout (provider.py:85-94)."""
    for sub in ("meta", "pointcloud1", "pointcloud2", "split"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    files = {}
    for i in range(n):
        npts1, npts2 = int(rng.integers(20, 60)), int(rng.integers(20, 60))
        if i == 3:
            npts2 = 0  # empty cloud branch (provider.py:95-98)
        pc1 = rng.normal(size=(npts1, 3)).astype(np.float32) * 2 + 5
        pc2 = rng.normal(size=(npts2, 3)).astype(np.float32) * 2 + 5
        tr = rng.normal(size=3)
        meta = {
            "translation": "\n".join("%.18e" % v for v in tr) + "\n",
            "rel_angle": float(rng.uniform(-1, 1)),
            "start_position": "\n".join("%.18e" % v for v in rng.normal(size=3) * 5) + "\n",
            "end_position": "\n".join("%.18e" % v for v in rng.normal(size=3) * 5) + "\n",
            "start_angle": float(rng.uniform(-3, 3)), "end_angle": float(rng.uniform(-3, 3)),
        }
        json.dump(meta, open(os.path.join(root, "meta", "%08d.json" % i), "w"))
        np.save(os.path.join(root, "pointcloud1", "%08d.npy" % i), pc1)
        np.save(os.path.join(root, "pointcloud2", "%08d.npy" % i), pc2)
        files[i] = dict(meta=meta, pc1=pc1, pc2=pc2)
    open(os.path.join(root, "split", "train.txt"), "w").write("\n".join(str(i) for i in (0, 1, 2, 4, 5)) + "\n")
    open(os.path.join(root, "split", "val.txt"), "w").write("\n".join(str(i) for i in (3, 6)) + "\n")
    return files


def make_eval_dataset(root, kind, n, rng):
    """Only what evaluation.evaluate reads: meta/<id>.json (evaluation.py:157, 215).  kind "kitti": tracklet ids, sequence and
    frame pairs, so that the test/val split by track id (:158-159) and the velocity tracks (:213-228, 82-110) are exercised."""
    os.makedirs(os.path.join(root, "meta"), exist_ok=True)
    metas = []
    for i in range(n):
        m = {"rel_angle": float(rng.uniform(-1, 1))}
        if kind == "kitti":
            track = int(rng.choice([1, 2, 3, 6, 9, 10]))
            frame = int(rng.integers(1, 14))
            m.update({"trackids": [track, track], "seq": int(rng.integers(0, 3)), "frames": [frame - 1, frame]})
            if i % 7 == 0:
                del m["seq"]            # a sample outside any track
        json.dump(m, open(os.path.join(root, "meta", "%08d.json" % i), "w"))
        metas.append(m)
    return metas


class _LegacyRaggedNumpy:
    """evaluation.process_velocities (evaluation.py:99) builds `np.array([(vec3, dt), ...])`.  The NumPy of the reference's era
    (< 1.24) made an object array of shape (n, 2) out of that ragged list; NumPy 2 raises.  For the fixture run only, the
    reference module's `np` is replaced by this proxy, which retries such a call with dtype=object (same values, same shape) and
    forwards everything else untouched."""

    def __init__(self, real):
        self._np = real

    def __getattr__(self, k):
        return getattr(self._np, k)

    def array(self, obj, *a, **k):
        try:
            return self._np.array(obj, *a, **k)
        except ValueError:
            return self._np.array(obj, dtype=object)


def eval_fixtures(tmp, evaluation, config):
    """evaluation.evaluate of the REFERENCE (evaluation.py:128-289) on synthetic predictions: the numbers of eval.json /
    eval_180.json and the velocity track files.  Base paths are named so that `is_test` is bound (:158-161)."""
    rng = np.random.default_rng(777)
    out, meta_out = {}, {}
    cfg = config.configGlobal
    evaluation.np = _LegacyRaggedNumpy(np)
    for kind, dirname, n in (("synth", "SynthEvalSet", 1012), ("kitti", "KITTI_tracklets_EvalSet", 60)):
        root = os.path.join(tmp, dirname)
        metas = make_eval_dataset(root, kind, n, rng)
        vars(cfg.data)["basepath"] = root
        val = list(range(n))
        gt_t = rng.normal(size=(n, 3)) * 0.5
        # a mix of very good, fair and poor predictions so that all three accuracy levels are populated
        scale = rng.choice([0.01, 0.06, 0.15, 0.6], size=(n, 1))
        pt = gt_t + rng.normal(size=(n, 3)) * scale
        gt_a = rng.uniform(-np.pi, np.pi, size=(n, 1))
        pa = gt_a + rng.normal(size=(n, 1)) * rng.choice([0.005, 0.05, 0.12, 1.0], size=(n, 1)) + np.pi * (rng.uniform(size=(n, 1)) < 0.2)
        gt_c = rng.normal(size=(n, 3)) * np.array([9.0, 9.0, 0.3])      # centroid distances 0 .. 30 m: all range buckets
        pc = gt_c + rng.normal(size=(n, 3)) * 0.2
        pt[5] = gt_t[5] + 20000.0                                       # dist_transl > 10000: skipped (:166-167)
        for k, v in (("val", np.asarray(val)), ("pred_t", pt), ("pred_a", pa), ("gt_t", gt_t), ("gt_a", gt_a), ("pred_c", pc), ("gt_c", gt_c)):
            out["%s_%s" % (kind, k)] = v
        meta_out[kind + "_meta"] = metas
        meta_out[kind + "_dirname"] = dirname
        for inv in (False, True):
            ed = os.path.join(tmp, "evalout_%s_%d" % (kind, inv))
            res, detail = evaluation.evaluate(cfg, val, pt, pa, gt_t, gt_a, pc, gt_c, eval_dir=ed, accept_inverted_angle=inv, detailed_eval=True,
                                              mean_time=0.25)
            written = json.load(open(os.path.join(ed, "eval_180.json" if inv else "eval.json")))
            assert written == evaluation.ns_to_dict(res)
            meta_out["%s_eval_%d" % (kind, inv)] = written
            out["%s_detail_levels_%d" % (kind, inv)] = np.array([d[0] for d in detail])
            out["%s_detail_dists_%d" % (kind, inv)] = np.array([[d[1], d[2]] for d in detail])
            vdir = os.path.join(ed, "velocities")
            tracks = {}
            if os.path.isdir(vdir):
                for f in sorted(os.listdir(vdir)):
                    tracks[f] = [float(x) for x in open(os.path.join(vdir, f)).read().split()]
            meta_out["%s_velocity_files_%d" % (kind, inv)] = tracks
    np.savez_compressed(os.path.join(HERE, "eval_vectors.npz"), **out)
    json.dump(meta_out, open(os.path.join(HERE, "eval_vectors.json"), "w"), indent=1, sort_keys=True)
    print("eval fixtures:", len(out), "arrays,", sum(len(meta_out[k]) for k in meta_out if "velocity" in k), "velocity files")


def main():
    stub_modules()
    sys.path.insert(0, os.path.join(REF, "tp_utils"))
    sys.path.insert(0, REF)
    rng = np.random.default_rng(20240917)
    out, meta_out = {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        root = os.path.join(tmp, "TinySet")
        files = make_dataset(root, rng)
        user_cfg = {
            "data": {"basepath": root},
            "model": {"num_points": 32, "options": {"early_stage_factor": 0.25}, "angles": {"num_bins": 10, "accept_inverted_angle": True}},
            "logging": {"basedir": os.path.join(tmp, "logs")},
            "training": {"batch_size": 4, "learning_rate": 0.002, "lr_extension": {"step": 7}},
            "evaluation": {"special": {"mode": "timings"}},
        }
        cfg_path = os.path.join(tmp, "TinyRun.json")
        json.dump(user_cfg, open(cfg_path, "w"))
        import provider  # must precede config (circular import)
        import config
        config.load_config(cfg_path)
        merged = {}
        config.namespace_to_dict(config.configGlobal, merged)
        save_path = os.path.join(tmp, "saved.json")
        config.save_config(save_path)
        meta_out["config_user"] = user_cfg
        meta_out["config_merged"] = json.loads(json.dumps(merged).replace(tmp, "<TMP>"))
        meta_out["config_saved_equals_merged"] = json.load(open(save_path)) == merged
        meta_out["tmp_token"] = "<TMP>"
        # dataset as fixture input
        for i, f in files.items():
            out[f"ds_pc1_{i}"] = f["pc1"]
            out[f"ds_pc2_{i}"] = f["pc2"]
        meta_out["ds_meta"] = {str(i): f["meta"] for i, f in files.items()}
        meta_out["ds_train"] = provider.getDataFiles(os.path.join(root, "split", "train.txt"))
        meta_out["ds_val"] = provider.getDataFiles(os.path.join(root, "split", "val.txt"))
        # provider.load_batch under a fixed legacy seed (np.random.choice call order matters: provider.py:97-98)
        np.random.seed(1234)
        b = provider.load_batch([0, 1, 2, 3])
        for k, v in zip(("pcs1", "pcs2", "translations", "rel_angles", "pc1centers", "pc2centers", "pc1angles", "pc2angles"), b):
            out["lb_" + k] = v
        np.random.seed(99)
        b2 = provider.load_batch([6, 5], override_batch_size=3)   # padded batch: row 2 is np.empty garbage (quirk A6 iv)
        for k, v in zip(("pcs1", "pcs2", "translations", "rel_angles", "pc1centers", "pc2centers", "pc1angles", "pc2angles"), b2):
            out["lb2_" + k] = v[:2]
        b3 = provider.load_batch([4, 0], override_batch_size=2, dont_load_pointclouds=True)
        out["lb3_translations"] = b3[2]
        out["lb3_pc2angles"] = b3[7]
        np.random.seed(7)
        x = rng.normal(size=(2, 5, 3))
        out["jit_in"] = x
        out["jit_out"] = provider.jitter_point_cloud(x.copy())

        # models/tp8.py NumPy decode (cfg.model.angles.num_bins = 10 from the merged config)
        import importlib
        tp8 = importlib.import_module("models.tp8")
        lg = rng.normal(size=(9, 20)).astype(np.float32) * 2
        lg[4, :10] = 0.0
        lg[4, 2] = lg[4, 7] = 1.5  # tie -> first arg-max
        out["dec_logits"] = lg
        out["dec_angles"] = tp8.classLogits2angle(lg)
        out["dec_class2angle"] = np.array([tp8.class2angle(c, r) for c, r in ((0, 0.1), (5, 0.3), (9, 0.9), (7, -0.2))])

        import evaluation
        ts, gts = rng.normal(size=(6, 3)) * 0.1, rng.normal(size=(6, 3)) * 0.1
        an, gan = rng.uniform(-3.2, 3.2, 6), rng.uniform(-3.2, 3.2, 6)
        out["ev_t"], out["ev_gt_t"], out["ev_a"], out["ev_gt_a"] = ts, gts, an, gan
        out["ev_transl_dist"] = np.array([evaluation.eval_translation(t, g)[0] for t, g in zip(ts, gts)])
        out["ev_transl_lvl"] = np.array([evaluation.eval_translation(t, g)[1] for t, g in zip(ts, gts)])
        for inv in (False, True):
            out[f"ev_angle_dist_{int(inv)}"] = np.array([evaluation.eval_angle(a, g, inv)[0] for a, g in zip(an, gan)])
            out[f"ev_angle_lvl_{int(inv)}"] = np.array([evaluation.eval_angle(a, g, inv)[1] for a, g in zip(an, gan)])
            out[f"ev_transform_{int(inv)}"] = np.array([evaluation.eval_transform(t, g, a, ga, inv) for t, g, a, ga in zip(ts, gts, an, gan)])

        import pointcloud
        arr = rng.normal(size=(4, 3))
        s = pointcloud.np_to_str(arr)
        meta_out["np_to_str"] = s
        out["str_to_np"] = pointcloud.str_to_np(s)
        out["np_to_str_in"] = arr
        out["mat_angle"] = pointcloud.get_mat_angle(np.array([1.0, -2.0, 0.5]), 0.7, rotation_center=np.array([3.0, 4.0, 0.0]))
        ctr, gctr = rng.normal(size=(5, 3)), rng.normal(size=(5, 3))
        pt, pa = rng.normal(size=(5, 3)), rng.uniform(-3, 3, 5)
        out["ttc_in_t"], out["ttc_in_a"], out["ttc_in_c"], out["ttc_in_g"] = pt, pa, ctr, gctr
        out["ttc_out"] = pointcloud.translate_transform_to_new_center_of_rotation(pt, pa, ctr, gctr)

        eval_fixtures(tmp, evaluation, config)

    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    json.dump(meta_out, open(os.path.join(HERE, "reference_vectors.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
