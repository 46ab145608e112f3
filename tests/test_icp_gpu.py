"""GPU: ICP refinement (alignnet_icp_refine*) against oracle/icp_ref.py -- the restatement of the reference's
icp.icp_p2point (icp.py:69-78, Open3D registration_icp, z-constrained point-to-point) used by train.py:463-484.
Bar: both sides compute in fp64 on fp32 points and make the same nearest-neighbour decisions, so the transforms agree to
1e-9 and fitness / rmse / iteration counts exactly (rmse to 1e-12); a known motion is recovered to 1e-6."""
import numpy as np
import pytest

import alignnet3d
from oracle import icp_ref as I
from tests.helpers import small_cfg

pytestmark = pytest.mark.gpu


def _pairs(n_pairs, seed, big=False):
    rng = np.random.default_rng(seed)
    src, dst, inits, truth = [], [], [], []
    for k in range(n_pairs):
        n2 = int(rng.integers(300, 900)) if not big else 14000
        q = (rng.uniform(-1, 1, (n2, 3)) * [2.2, 0.9, 0.7] + rng.uniform(-15, 15, 3)).astype(np.float32)
        th, t = rng.uniform(-0.08, 0.08), rng.uniform(-0.06, 0.06, 3)
        R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
        keep = rng.permutation(n2)[: int(n2 * 0.7)]
        p = ((q[keep].astype(np.float64) - t) @ R).astype(np.float32)          # q = R p + t
        if k % 3 == 1:
            p = p + rng.normal(0, 0.004, p.shape).astype(np.float32)          # noisy copy: no exact fixed point
        src.append(p); dst.append(q)
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
        truth.append(T)
        # start from the truth disturbed by a small motion about the target's centre (what the network's prediction is)
        inits.append(I.get_mat_angle(rng.normal(0, 0.01, 3), rng.normal(0, 0.01), rotation_center=q.mean(0)) @ T)
    return src, dst, inits, truth


def test_icp_matches_oracle(gpu_required):
    eng = alignnet3d.Engine(small_cfg(N=64, nb=12))
    src, dst, inits, truth = _pairs(7, seed=2)
    src.append(np.zeros((0, 3), np.float32)); dst.append(dst[0]); inits.append(np.eye(4)); truth.append(np.eye(4))   # empty source
    src.append(src[0]); dst.append(np.zeros((0, 3), np.float32)); inits.append(inits[1]); truth.append(np.eye(4))    # empty target
    for radius, its in ((0.1, 30), (0.25, 3), (0.1, 0)):
        res = eng.icp_refine(src, dst, inits, radius=radius, its=its)
        for k in range(len(src)):
            T, fit, rmse, it = I.icp_p2point_z(src[k], dst[k], inits[k], radius, its)
            np.testing.assert_allclose(res["transforms"][k], T, rtol=0, atol=1e-9, err_msg="pair %d" % k)
            assert res["fitness"][k] == fit and abs(res["rmse"][k] - rmse) < 1e-12 and res["iterations"][k] == it, (k, res["iterations"][k], it)
    # exact copies under a small motion are recovered
    res = eng.icp_refine(src[:7], dst[:7], inits[:7], radius=0.25, its=50)
    for k in (0, 2, 3, 5, 6):
        np.testing.assert_allclose(res["transforms"][k], truth[k], rtol=0, atol=1e-6)
        assert res["fitness"][k] == 1.0 and res["rmse"][k] < 1e-6
    with pytest.raises(RuntimeError):
        eng.icp_refine(src[:1], dst[:1], inits[:1], radius=0.0)
    with pytest.raises(RuntimeError):
        eng.icp_refine_rows([0], inits[:1])            # no dataset uploaded
    eng.close()


def test_icp_rows_and_large_cloud(gpu_required):
    """Dataset-resident clouds addressed by rows, and a target cloud larger than the LDS stage (tail served from L2)."""
    eng = alignnet3d.Engine(small_cfg(N=64, nb=12))
    src, dst, inits, _ = _pairs(4, seed=5)
    bs, bd, bi, _ = _pairs(1, seed=6, big=True)
    src += bs; dst += bd; inits += bi
    off = np.zeros((len(src) + 1, 2), np.int64)
    off[1:, 0] = np.cumsum([len(s) for s in src]); off[1:, 1] = np.cumsum([len(t) for t in dst])
    eng.upload_dataset(np.concatenate(src), np.concatenate(dst), off, np.zeros((len(src), 12), np.float32))
    rows = [4, 1, 3, 1]
    res = eng.icp_refine_rows(rows, [inits[r] for r in rows], radius=0.1, its=10)
    direct = eng.icp_refine([src[r] for r in rows], [dst[r] for r in rows], [inits[r] for r in rows], radius=0.1, its=10)
    assert np.array_equal(res["transforms"], direct["transforms"]) and np.array_equal(res["iterations"], direct["iterations"])
    T, fit, rmse, it = I.icp_p2point_z(src[4], dst[4], inits[4], 0.1, 10)
    np.testing.assert_allclose(res["transforms"][0], T, rtol=0, atol=1e-9)
    assert res["fitness"][0] == fit and res["iterations"][0] == it
    eng.close()


@pytest.mark.parametrize("offset", [0.0, 512.0, 4096.0])
def test_icp_exact_ties_and_far_frames(gpu_required, offset):
    """The nearest-neighbour scan is certified in fp32 before it is decided in fp64 (alignnet_icp.hip: icp_prefilter_eps); this is the case it must not lose:
    a lattice target (spacing 2^-4, exactly representable also `offset` metres from the origin, where an fp32 coordinate carries 2^-15 .. 2^-12 of absolute
    rounding -- three to four orders above the 1e-9 the transforms are held to) and source points on the cell centres of its faces, so that every source
    point has FOUR exactly equidistant nearest targets in the first evaluation (the oracle's argmin takes the first index); plus duplicated target points
    (exact ties inside one lane's slice and across lanes).  Far from the origin ONE estimate is compared (its = 1: the transform is a function of the first
    evaluation's correspondences alone, i.e. of the tie-breaking) -- from the second evaluation on, the four candidates of a source point differ by fractions
    of an ulp of the transformed point, and which one wins depends on whether T p was formed with fused multiply-adds (the kernel) or not (NumPy): the all-fp64
    kernel of the commit before disagrees with the oracle there in the same way.  At the origin the whole iteration is compared."""
    eng = alignnet3d.Engine(small_cfg(N=64, nb=12))
    g = np.arange(12, dtype=np.float64) / 16.0
    lat = np.stack(np.meshgrid(g, g, g[:5], indexing="ij"), -1).reshape(-1, 3) + offset            # 720 lattice points
    rng = np.random.default_rng(3)
    lat = lat[rng.permutation(len(lat))]                                                            # index order unrelated to position
    dst = np.concatenate([lat, lat[:97]]).astype(np.float32)                                        # + 97 duplicates
    assert np.array_equal(dst.astype(np.float64)[: len(lat)], lat)                                  # exactly representable
    src = (lat[(lat[:, 0] < offset + 10 / 16) & (lat[:, 1] < offset + 10 / 16)] + [1 / 32, 1 / 32, 0.0]).astype(np.float32)
    inits = [np.eye(4)]   # (exact T p: the ties of the first evaluation are exact)
    if offset == 0.0:
        inits.append(I.get_mat_angle(np.array([1e-3, -2e-3, 0.0]), 5e-4, rotation_center=dst.mean(0).astype(np.float64)))
    for init in inits:
        for radius, its in (((0.1, 30), (0.05, 5)) if offset == 0.0 else ((0.1, 1), (0.05, 1))):
            res = eng.icp_refine([src], [dst], [init], radius=radius, its=its)
            T, fit, rmse, it = I.icp_p2point_z(src, dst, init, radius, its)
            np.testing.assert_allclose(res["transforms"][0], T, rtol=0, atol=1e-9 * max(1.0, offset))
            rtol = 1e-12 if offset == 0.0 else 1e-9   # (far frames: the second evaluation's distances carry ulp(offset) = 1e-13 .. 1e-12 of the transformed points' rounding)
            assert res["fitness"][0] == fit and abs(res["rmse"][0] - rmse) < rtol and res["iterations"][0] == it, (offset, res["iterations"][0], it, res["fitness"][0], fit)
    eng.close()
