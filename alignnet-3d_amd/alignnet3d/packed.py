"""Packed dataset cache for the batch loader.

The reference loader opens three files per example per load and parses a JSON document plus four
np.loadtxt strings (provider.py:85-94); at >100 k pairs/s on the GPU that Python loop is the wall-clock
bottleneck by orders of magnitude.  `pack_dataset` converts a dataset directory once into
    <cache>/points1.npy, points2.npy   all clouds concatenated ([sum n, 3], original dtype)
    <cache>/offsets.npy                [n_examples + 1, 2] row offsets into the two blobs
    <cache>/labels.npy                 [n_examples, 12] float64: translation(3) rel_angle start_position(3)
                                       end_position(3) start_angle end_angle
    <cache>/ids.npy                    example ids (the integers of split/*.txt)
`PackedDataset.load_batch` returns exactly the reference's 8-tuple (float64, np.empty padding rows) and
consumes np.random in the same order (`choice(n1, N)` then `choice(n2, N)` per example), so a seeded run is
bit-identical to the file-based loader (tests/test_surface.py checks it against reference-generated batches)."""
import json
import os

import numpy as np


def _loadtxt(s):
    return np.array([float(t) for t in s.split()], dtype=np.float64)


def dataset_fingerprint(basepath):
    """What a cache is valid for: the example ids present and the newest modification time of the three per-example
    directories (a re-generated or extended dataset changes one of them)."""
    names = sorted(f for f in os.listdir(os.path.join(basepath, "meta")) if f.endswith(".json"))
    mtime = max([os.stat(os.path.join(basepath, d)).st_mtime_ns for d in ("meta", "pointcloud1", "pointcloud2")] +
                [os.stat(os.path.join(basepath, "meta", f)).st_mtime_ns for f in names[:1] + names[-1:]])
    return {"n_examples": len(names), "first": names[0] if names else "", "last": names[-1] if names else "", "mtime_ns": int(mtime)}


def cache_is_current(basepath, cache_dir):
    try:
        with open(os.path.join(cache_dir, "manifest.json")) as fh:
            return json.load(fh) == dataset_fingerprint(basepath)
    except (OSError, ValueError):
        return False


def pack_dataset(basepath, cache_dir=None):
    """Build the cache.  Every table is written to a temporary name and renamed into place, the manifest last, so a reader
    never maps a half-written file and an interrupted build is not mistaken for a cache."""
    cache_dir = cache_dir or os.path.join(basepath, "packed_cache")
    os.makedirs(cache_dir, exist_ok=True)
    fingerprint = dataset_fingerprint(basepath)
    try:
        os.remove(os.path.join(cache_dir, "manifest.json"))
    except OSError:
        pass

    def save(name, arr):
        tmp = os.path.join(cache_dir, ".%s.%d.tmp.npy" % (name, os.getpid()))
        np.save(tmp, arr)
        os.replace(tmp, os.path.join(cache_dir, name))
    ids = sorted(int(f[:-5]) for f in os.listdir(os.path.join(basepath, "meta")) if f.endswith(".json"))
    labels = np.empty((len(ids), 12), np.float64)
    offs = np.zeros((len(ids) + 1, 2), np.int64)
    blobs = ([], [])
    for row, ex in enumerate(ids):
        stem = str(ex).zfill(8)
        with open(os.path.join(basepath, "meta", stem + ".json")) as fh:
            m = json.load(fh)
        labels[row, 0:3] = _loadtxt(m["translation"])
        labels[row, 3] = m["rel_angle"]
        labels[row, 4:7] = _loadtxt(m["start_position"])
        labels[row, 7:10] = _loadtxt(m["end_position"])
        labels[row, 10], labels[row, 11] = m["start_angle"], m["end_angle"]
        for t, sub in enumerate(("pointcloud1", "pointcloud2")):
            pc = np.load(os.path.join(basepath, sub, stem + ".npy"))
            pc = pc[:, :3] if pc.ndim == 2 and pc.shape[0] > 0 else np.zeros((0, 3), np.float32)
            blobs[t].append(pc)
            offs[row + 1, t] = offs[row, t] + pc.shape[0]
    for t, name in enumerate(("points1.npy", "points2.npy")):
        dt = np.result_type(*[b.dtype for b in blobs[t]]) if blobs[t] else np.float32
        arr = np.concatenate([b.astype(dt, copy=False) for b in blobs[t]], axis=0) if blobs[t] else np.zeros((0, 3), dt)
        save(name, arr)
    save("offsets.npy", offs)
    save("labels.npy", labels)
    save("ids.npy", np.asarray(ids, np.int64))
    tmp = os.path.join(cache_dir, ".manifest.%d.tmp" % os.getpid())
    with open(tmp, "w") as fh:
        json.dump(fingerprint, fh)
    os.replace(tmp, os.path.join(cache_dir, "manifest.json"))
    return cache_dir


class PackedDataset:
    def __init__(self, cache_dir):
        self.p = [np.load(os.path.join(cache_dir, "points%d.npy" % t), mmap_mode="r") for t in (1, 2)]
        self.off = np.load(os.path.join(cache_dir, "offsets.npy"))
        self.labels = np.load(os.path.join(cache_dir, "labels.npy"))
        ids = np.load(os.path.join(cache_dir, "ids.npy"))
        self.row = {int(e): i for i, e in enumerate(ids)}

    def __len__(self):
        return len(self.row)

    def rows_of(self, indices):
        """Example ids (split/*.txt integers) -> rows of the packed tables."""
        return np.asarray([self.row[int(e)] for e in indices], np.int32)

    def upload(self, engine):
        """Copy the whole dataset into HBM once (Engine.upload_dataset); batches are then drawn on the device."""
        engine.upload_dataset(np.asarray(self.p[0])[:, :3], np.asarray(self.p[1])[:, :3], self.off, self.labels)

    def load_batch(self, indices, num_points, batch_size, num_channels=3, dont_load_pointclouds=False, on_empty=None):
        B, N = batch_size, num_points
        pcs1, pcs2 = np.empty((B, N, num_channels)), np.empty((B, N, num_channels))
        translations, rel_angles = np.empty((B, 3)), np.empty((B, 1))
        pc1centers, pc2centers = np.empty((B, 3)), np.empty((B, 3))
        pc1angles, pc2angles = np.empty((B, 1)), np.empty((B, 1))
        for r, ex in enumerate(indices):
            i = self.row[int(ex)]
            if not dont_load_pointclouds:
                for t, dst in ((0, pcs1), (1, pcs2)):
                    lo, hi = self.off[i, t], self.off[i + 1, t]
                    n = int(hi - lo)
                    if n > 0:   # same RNG consumption as provider.py:97-98: one choice() per non-empty cloud, pc1 first
                        dst[r] = self.p[t][lo + np.random.choice(n, N, replace=True)]
                    else:
                        dst[r] = 0.0
                if on_empty is not None and (self.off[i + 1, 0] == self.off[i, 0] or self.off[i + 1, 1] == self.off[i, 1]):
                    on_empty(ex)
            L = self.labels[i]
            translations[r], rel_angles[r] = L[0:3], L[3]
            pc1centers[r], pc2centers[r] = L[4:7], L[7:10]
            pc1angles[r], pc2angles[r] = L[10], L[11]
        return pcs1, pcs2, translations, rel_angles, pc1centers, pc2centers, pc1angles, pc2angles
