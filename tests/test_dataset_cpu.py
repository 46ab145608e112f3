"""CPU: the sampler restatement (oracle/dataset_ref.py) has the reference loader's distribution-level properties
(provider.py:97-98 sampling with replacement from the example's own cloud; provider.py:60-71 clipped jitter) and the
packed tables feed it the same examples the file-based loader reads."""
import numpy as np

from oracle import dataset_ref as D


def _toy(n=6, seed=1):
    rng = np.random.default_rng(seed)
    cnt = rng.integers(1, 40, (n, 2)); cnt[3, 0] = 0
    off = np.zeros((n + 1, 2), np.int64); off[1:] = np.cumsum(cnt, 0)
    pts = [rng.normal(size=(off[-1, t], 3)).astype(np.float32) for t in range(2)]
    return pts, off, rng.normal(size=(n, 12)).astype(np.float32)


def test_oracle_sampler_properties():
    pts, off, lab = _toy()
    rows = [0, 3, 5, 3]
    a, b, labs, picks = D.sample_batch(pts, off, lab, rows, 128, seed=42)
    for r, row in enumerate(rows):
        for t, arr in ((0, a), (1, b)):
            lo, hi = off[row, t], off[row + 1, t]
            if hi == lo:
                assert not arr[r].any()
                continue
            assert picks[r, t].min() >= 0 and picks[r, t].max() < hi - lo
            assert np.array_equal(arr[r], pts[t][lo + picks[r, t]])
    assert np.array_equal(a[1], a[3]) and np.array_equal(b[1], b[3])          # a function of (seed, row, tower, point) only
    assert np.array_equal(labs["translations"], lab[rows, 0:3]) and np.array_equal(labs["pc2_angles"], lab[rows, 11:12])
    a2 = D.sample_batch(pts, off, lab, rows, 128, seed=43)[0]
    assert not np.array_equal(a2, a)
    j = D.sample_batch(pts, off, lab, rows, 128, seed=42, sigma=0.01, clip=0.02)[0]
    assert np.abs(j - a).max() <= 0.02 + 1e-7 and np.abs(j - a)[0].max() > 0
    big = D.sample_batch(pts, off, lab, [0] * 1, 20000, seed=7)[3][0, 0]
    cnt = np.bincount(big, minlength=int(off[1, 0] - off[0, 0]))
    expect = 20000 / cnt.size
    assert ((cnt - expect) ** 2 / expect).sum() < 3 * cnt.size          # loose chi-square


def test_icp_oracle_recovers_motion_and_matches_kdtree():
    """oracle/icp_ref.py: the z-constrained ICP recovers a known planar motion; its brute-force correspondences equal a
    KD-tree's (scipy), which is what Open3D uses."""
    from scipy.spatial import cKDTree
    from oracle import icp_ref as I
    rng = np.random.default_rng(0)
    q = rng.uniform(-1, 1, (500, 3)) * [2.0, 0.9, 0.7]
    th, t = 0.06, np.array([0.04, -0.03, 0.02])
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    p = (q[rng.permutation(500)[:350]] - t) @ R
    T, fit, rmse, k = I.icp_p2point_z(p, q, None, radius=0.25, its=50)
    assert fit == 1.0 and rmse < 1e-9 and k < 50
    np.testing.assert_allclose(T[:3, :3], R, atol=1e-9); np.testing.assert_allclose(T[:3, 3], t, atol=1e-9)
    tr, ang = I.transform_to_prediction(T)
    assert abs(ang - th) < 1e-9 and np.allclose(tr, t)
    pp, qq, f, r = I._evaluate(p, q, np.eye(4), 0.1)
    d, j = cKDTree(q).query(p, distance_upper_bound=0.1)
    assert np.array_equal(qq, q[j[np.isfinite(d)]]) and abs(f - np.isfinite(d).mean()) < 1e-15
    M = I.get_mat_angle([1.0, 2.0, 3.0], 0.3, rotation_center=[4.0, 5.0, 6.0])     # tp_utils/pointcloud.py:279-289
    c = np.array([4.0, 5.0, 6.0, 1.0])
    np.testing.assert_allclose(M @ c, c + [1.0, 2.0, 3.0, 0.0], atol=1e-12)       # the centre only translates
    assert I.icp_p2point_z(np.zeros((0, 3)), q, None)[3] == 0                       # empty source: init returned
