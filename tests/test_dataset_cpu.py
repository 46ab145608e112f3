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
