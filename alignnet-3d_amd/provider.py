"""Batch loader of the drop-in (counterpart of the reference's provider.py:60-136): per-example
`meta/%08d.json` + `pointcloud{1,2}/%08d.npy`, re-sampled WITH replacement to cfg.model.num_points at every
load (also in eval: quirk A6(v)), float64 arrays, the same 8-tuple, and the same np.random call order so that a
seeded run reproduces the reference's batches bit for bit (tests/golden)."""
import io
import json
import logging

import numpy as np

from config import configGlobal as cfg

logger = logging.getLogger("tp")


def str_to_np(s):
    """Plain-text array as written by np.savetxt (tp_utils/pointcloud.py:260-265, plaintext branch)."""
    return np.loadtxt(io.BytesIO(s.encode("ascii")))


def getDataFiles(list_filename):
    return [int(line.rstrip()) for line in open(list_filename)]


def jitter_point_cloud(batch_data, sigma=0.01, clip=0.05):
    """Per-point clipped Gaussian jitter (provider.py:60-71)."""
    B, N, C = batch_data.shape
    assert clip > 0
    noise = np.clip(sigma * np.random.randn(B, N, C), -1 * clip, clip)
    noise += batch_data
    return noise


def _resample(pc, n):
    if pc.shape[0] > 0:
        return pc[np.random.choice(pc.shape[0], n, replace=True), :]
    return np.zeros((n, 3), dtype=np.float32)


def load_from_separate_files(idx, dont_load_pointclouds=False):
    stem = str(idx).zfill(8)
    with open("%s/meta/%s.json" % (cfg.data.basepath, stem)) as fh:
        meta = json.load(fh)
    labels = (str_to_np(meta["translation"]), meta["rel_angle"], str_to_np(meta["start_position"]),
              str_to_np(meta["end_position"]), meta["start_angle"], meta["end_angle"])
    if dont_load_pointclouds:
        return labels
    pc1 = np.load("%s/pointcloud1/%s.npy" % (cfg.data.basepath, stem))
    pc2 = np.load("%s/pointcloud2/%s.npy" % (cfg.data.basepath, stem))
    if pc1.shape[0] == 0 or pc2.shape[0] == 0:
        logger.error("Empty pointcloud! %s" % idx)
    n = cfg.model.num_points
    pc1 = _resample(pc1, n)   # pc1 first, then pc2: the RNG stream order is part of the contract
    pc2 = _resample(pc2, n)
    return (pc1, pc2) + labels


_packed = None


def use_packed_cache(cache_dir=None, dist=None):
    """Switch load_batch to the packed cache (alignnet3d/packed.py).  Same tuple, same np.random call order, bit-identical
    batches.  The cache is (re)built when it is missing or its manifest no longer matches the dataset (example count, newest
    modification time) -- by rank 0 only in a multi-process run (RANK / `dist`), the other ranks wait at a barrier."""
    global _packed
    import os
    from alignnet3d.packed import PackedDataset, pack_dataset, cache_is_current
    cache_dir = cache_dir or os.path.join(cfg.data.basepath, "packed_cache")
    rank = int(os.environ.get("RANK", "0"))
    if dist is None and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as td
        dist = td if td.is_initialized() else None
    if rank == 0 and not cache_is_current(cfg.data.basepath, cache_dir):
        pack_dataset(cfg.data.basepath, cache_dir)
    if dist is not None:
        dist.barrier()
    elif rank != 0:
        raise RuntimeError("use_packed_cache: a multi-process run needs an initialised process group (rank 0 builds the cache, "
                           "the others wait for it)")
    _packed = PackedDataset(cache_dir)
    return _packed


def load_batch(indices, override_batch_size=None, dont_load_pointclouds=False):
    """Returns (pcs1, pcs2, translations, rel_angles, pc1centers, pc2centers, pc1angles, pc2angles), float64.
    Rows past len(indices) are left uninitialised exactly like the reference's np.empty (quirk A6(iv)); the
    engine itself accepts any batch size, so callers may simply slice them off."""
    B = cfg.training.batch_size if override_batch_size is None else override_batch_size
    N, C = cfg.model.num_points, cfg.data.num_channels
    if _packed is not None:
        return _packed.load_batch(indices, N, B, C, dont_load_pointclouds, on_empty=lambda ex: logger.error("Empty pointcloud! %s" % ex))
    pcs1, pcs2 = np.empty((B, N, C)), np.empty((B, N, C))
    translations, rel_angles = np.empty((B, 3)), np.empty((B, 1))
    pc1centers, pc2centers = np.empty((B, 3)), np.empty((B, 3))
    pc1angles, pc2angles = np.empty((B, 1)), np.empty((B, 1))
    for row, ex in enumerate(indices):
        rec = load_from_separate_files(ex, dont_load_pointclouds=dont_load_pointclouds)
        if not dont_load_pointclouds:
            pcs1[row] = rec[0][:, :3]
            pcs2[row] = rec[1][:, :3]
            rec = rec[2:]
        translations[row], rel_angles[row] = rec[0], rec[1]
        pc1centers[row], pc2centers[row] = rec[2], rec[3]
        pc1angles[row], pc2angles[row] = rec[4], rec[5]
    return pcs1, pcs2, translations, rel_angles, pc1centers, pc2centers, pc1angles, pc2angles
