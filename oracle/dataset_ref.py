"""CPU restatement of the device-side batch sampler (alignnet-3d_amd/csrc/alignnet_dataset.hip).

TEST INFRASTRUCTURE ONLY.  The sampler has no counterpart in the reference beyond its *distribution*
(provider.py:97-98 `np.random.choice(n, N, replace=True)`, provider.py:60-71 `clip(sigma * randn, +-clip)`); this file
restates the engine's counter-hash stream in NumPy so that the gather indices can be checked bit for bit and the
jitter to float rounding.  parity unpinned against np.random by construction (documented in include/alignnet_hip.h).
"""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def mix64(x):
    x = np.asarray(x, np.uint64)
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint64(30)); x = x * np.uint64(0xBF58476D1CE4E5B9)
        x = x ^ (x >> np.uint64(27)); x = x * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def sample_batch(points, offsets, labels, rows, N, seed, sigma=0.0, clip=0.05):
    """points: (points1, points2) float32 [*, 3]; offsets [n+1, 2]; labels [n, 12]; rows: example rows.
    Returns pcs1, pcs2 [B, N, 3] float32 and the six label arrays, plus the picked point indices [B, 2, N]."""
    B = len(rows)
    out = [np.zeros((B, N, 3), np.float32), np.zeros((B, N, 3), np.float32)]
    picks = np.full((B, 2, N), -1, np.int64)
    n = np.arange(N, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for b, row in enumerate(rows):
            for t in range(2):
                lo = int(offsets[row, t]); cnt = int(offsets[row + 1, t]) - lo
                if cnt <= 0:
                    continue
                key = mix64(np.uint64(seed) ^ (np.uint64(row) * np.uint64(0x9E3779B97F4A7C15)) ^
                            ((np.uint64(2) * n + np.uint64(t)) * np.uint64(0xD1B54A32D192ED03)))
                pick = (((key >> np.uint64(32)) * np.uint64(cnt)) >> np.uint64(32)).astype(np.int64)
                picks[b, t] = pick
                v = np.asarray(points[t], np.float32)[lo + pick].copy()
                if sigma > 0:
                    k2 = mix64(key + np.uint64(0x632BE59BD9B4E019)); k3 = mix64(key + np.uint64(0xC6BC279692B5C323))
                    f = np.float32(1.0 / 16777216.0)
                    u0 = ((k2 >> np.uint64(40)).astype(np.float32) + np.float32(0.5)) * f
                    u1 = ((k2 >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float32) * f
                    u2 = ((k3 >> np.uint64(40)).astype(np.float32) + np.float32(0.5)) * f
                    u3 = ((k3 >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float32) * f
                    r0 = np.sqrt(np.float32(-2.0) * np.log(u0)); r1 = np.sqrt(np.float32(-2.0) * np.log(u2))
                    tp = np.float32(6.28318530718)
                    z = np.stack([r0 * np.cos(tp * u1), r0 * np.sin(tp * u1), r1 * np.cos(tp * u3)], 1).astype(np.float32)
                    v = v + np.clip(np.float32(sigma) * z, -np.float32(clip), np.float32(clip))
                out[t][b] = v
    lab = np.asarray(labels, np.float32)[np.asarray(rows)]
    labs = dict(translations=lab[:, 0:3], rel_angles=lab[:, 3:4], pc1_centers=lab[:, 4:7], pc2_centers=lab[:, 7:10],
                pc1_angles=lab[:, 10:11], pc2_angles=lab[:, 11:12])
    return out[0], out[1], labs, picks
