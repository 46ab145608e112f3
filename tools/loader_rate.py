# CPU: file-based reference-style loader vs packed cache (same seeded batches), 512 synthetic examples, N=1024
import json, os, sys, tempfile, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'alignnet-3d_amd')); sys.path.insert(0, ROOT)
from oracle import alignnet_ref as R
with tempfile.TemporaryDirectory() as tmp:
    root = os.path.join(tmp, "SynthBig"); n = 512
    d = R.synth_pairs(n, 1500, dtype=np.float32)
    for sub in ("meta", "pointcloud1", "pointcloud2", "split"): os.makedirs(os.path.join(root, sub))
    txt = lambda v: "\n".join("%.18e" % x for x in np.ravel(v)) + "\n"
    for i in range(n):
        json.dump({"translation": txt(d["translations"][i]), "rel_angle": float(d["rel_angles"][i, 0]), "start_position": txt(d["pc1_centers"][i]),
                   "end_position": txt(d["pc2_centers"][i]), "start_angle": float(d["pc1_angles"][i, 0]), "end_angle": float(d["pc2_angles"][i, 0])},
                  open(os.path.join(root, "meta", "%08d.json" % i), "w"))
        np.save(os.path.join(root, "pointcloud1", "%08d.npy" % i), d["pcs1"][i]); np.save(os.path.join(root, "pointcloud2", "%08d.npy" % i), d["pcs2"][i])
    for f in ("train", "val"): open(os.path.join(root, "split", f + ".txt"), "w").write("\n".join(map(str, range(n))) + "\n")
    cfgp = os.path.join(tmp, "c.json"); json.dump({"data": {"basepath": root}, "logging": {"basedir": tmp}, "model": {"num_points": 1024}, "training": {"batch_size": 256}}, open(cfgp, "w"))
    import config, provider
    config.load_config(cfgp)
    idx = list(range(256))
    np.random.seed(0); t = time.perf_counter(); a = provider.load_batch(idx); t_file = time.perf_counter() - t
    t = time.perf_counter(); provider.use_packed_cache(); t_pack = time.perf_counter() - t
    np.random.seed(0); t = time.perf_counter(); b = provider.load_batch(idx); t_packed = time.perf_counter() - t
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    print("file-based: %.1f pairs/s   packed: %.1f pairs/s (x%.1f)   one-off packing of %d examples: %.2f s" % (256 / t_file, 256 / t_packed, t_file / t_packed, n, t_pack))
