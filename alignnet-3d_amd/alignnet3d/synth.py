"""Synthetic point-cloud pairs for benchmarks and smoke runs (SURVEY.md 8(d) recipe: car-sized boxes at 4-20 m, the second
cloud the same object moved by <= 1 m and a yaw of U(-pi, pi) / 2, clipped Gaussian jitter as provider.py:60-71).
Input generation only -- no network arithmetic lives here."""
import numpy as np


def synth_pairs(B, N, seed=1234, dtype=np.float32):
    """Car-sized boxes at 4-20 m, second cloud = same object moved by <=1 m and a
    yaw of U(-pi,pi)/2, with clipped Gaussian jitter (mirrors provider.py:60-71 and
    tp_utils/pointcloud.py:522-534 in spirit; not reference data)."""
    rng = np.random.default_rng(seed)
    yaw1 = rng.uniform(-np.pi, np.pi, B)
    dist = rng.uniform(4, 20, B)
    bear = rng.uniform(-np.pi, np.pi, B)
    c1 = np.stack([dist * np.cos(bear), dist * np.sin(bear), np.zeros(B)], 1)
    tl = rng.uniform(0, 1, B)
    td = rng.uniform(-np.pi, np.pi, B)
    trans = np.stack([tl * np.cos(td), tl * np.sin(td), np.zeros(B)], 1)
    rel = rng.uniform(-np.pi, np.pi, B) / 2
    yaw2 = yaw1 + rel
    c2 = c1 + trans
    ext = np.array([4.5, 1.8, 1.5])

    def cloud(c, yaw):
        p = (rng.uniform(-0.5, 0.5, (B, N, 3))) * ext
        # push a random coordinate to the box surface ("surface-ish")
        ax = rng.integers(0, 3, (B, N))
        sgn = rng.choice([-0.5, 0.5], (B, N))
        p[np.arange(B)[:, None], np.arange(N)[None, :], ax] = sgn * ext[ax]
        cs, sn = np.cos(yaw), np.sin(yaw)
        x = p[..., 0] * cs[:, None] - p[..., 1] * sn[:, None]
        y = p[..., 0] * sn[:, None] + p[..., 1] * cs[:, None]
        q = np.stack([x, y, p[..., 2]], -1) + c[:, None, :]
        return q + np.clip(0.01 * rng.standard_normal((B, N, 3)), -0.05, 0.05)

    pcs1, pcs2 = cloud(c1, yaw1), cloud(c2, yaw2)
    wrap = lambda a: (a + np.pi) % (2 * np.pi) - np.pi
    return dict(
        pcs1=pcs1.astype(dtype), pcs2=pcs2.astype(dtype), translations=trans.astype(dtype),
        rel_angles=rel[:, None].astype(dtype), pc1_centers=c1.astype(dtype), pc2_centers=c2.astype(dtype),
        pc1_angles=wrap(yaw1)[:, None].astype(dtype), pc2_angles=wrap(yaw2)[:, None].astype(dtype))
