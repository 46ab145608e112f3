"""CPU restatement of the ICP refinement of the reference's evaluation (`icp.py:69-78` icp_p2point, called from
`train.py:463-484` with radius 0.1, `its` iterations, with_constraint=True, seeded by the network's prediction).

TEST INFRASTRUCTURE ONLY.  **parity unpinned**: the arithmetic lives in Open3D (`o3.registration_icp`,
`TransformationEstimationPointToPoint`), pinned by the reference only in prose (README.md:32, a private fork that adds
`with_constraint`), absent from /root/reference and not installable here.  Restated from Open3D's published algorithm
(Registration.cpp `RegistrationICP`, v0.7 line):
    T = init;  result = evaluate(T)                       # per source point: nearest target point within `radius`
    repeat max_iteration times:
        update = estimate(correspondences of result);  T = update @ T;  previous = result;  result = evaluate(T)
        stop when |fitness - previous.fitness| < 1e-6 and |inlier_rmse - previous.inlier_rmse| < 1e-6
    fitness = #correspondences / #source points,  inlier_rmse = sqrt(sum d^2 / #correspondences)
and, for the fork's `with_constraint=True` ("rotation about the z axis only", icp.py:76 / configs evaluation.special.icp),
the least-squares optimum of  sum |Rz(theta) p + t - q|^2 :
    theta = atan2(sum(p'_x q'_y - p'_y q'_x), sum(p'_x q'_x + p'_y q'_y)),   t = mean(q) - Rz(theta) mean(p)
with p', q' the centred correspondences (the planar Umeyama/Kabsch solution; z only translates).
`get_mat_angle` follows tp_utils/pointcloud.py:279-289.
"""
import numpy as np


def get_mat_angle(translation=None, rotation=None, rotation_center=(0.0, 0.0, 0.0)):
    c = np.asarray(rotation_center, np.float64)
    m1, m2, m3 = np.eye(4), np.eye(4), np.eye(4)
    m1[:3, 3] = -c
    m3[:3, 3] = c
    if translation is not None:
        m3[:3, 3] += np.asarray(translation, np.float64)
    if rotation is not None:
        a = float(rotation)
        m2[:3, :3] = [[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]]
    return m3 @ m2 @ m1


def _evaluate(src, dst, T, radius):
    p = src @ T[:3, :3].T + T[:3, 3]
    d2 = ((p[:, None, :] - dst[None, :, :]) ** 2).sum(-1)
    j = d2.argmin(1)                       # first index wins ties
    best = d2[np.arange(len(p)), j]
    ok = best <= radius * radius
    n = int(ok.sum())
    fitness = n / float(len(p)) if len(p) else 0.0
    rmse = float(np.sqrt(best[ok].sum() / n)) if n else 0.0
    return p[ok], dst[j[ok]], fitness, rmse


def _estimate_z(p, q):
    if len(p) == 0:
        return np.eye(4)
    mp, mq = p.mean(0), q.mean(0)
    pc, qc = p - mp, q - mq
    sxy = float((pc[:, 0] * qc[:, 1] - pc[:, 1] * qc[:, 0]).sum())
    sxx = float((pc[:, 0] * qc[:, 0] + pc[:, 1] * qc[:, 1]).sum())
    th = np.arctan2(sxy, sxx)
    U = np.eye(4)
    U[:3, :3] = [[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]]
    U[:3, 3] = mq - U[:3, :3] @ mp
    return U


def icp_p2point_z(src, dst, init=None, radius=0.1, its=30):
    """Returns (T [4,4] float64, fitness, inlier_rmse, iterations run)."""
    src, dst = np.asarray(src, np.float64)[:, :3], np.asarray(dst, np.float64)[:, :3]
    T = np.eye(4) if init is None else np.array(init, np.float64)
    if len(src) == 0 or len(dst) == 0:
        return T, 0.0, 0.0, 0
    p, q, fit, rmse = _evaluate(src, dst, T, radius)
    k = 0
    for k in range(1, its + 1):
        T = _estimate_z(p, q) @ T
        p, q, nfit, nrmse = _evaluate(src, dst, T, radius)
        done = abs(nfit - fit) < 1e-6 and abs(nrmse - rmse) < 1e-6
        fit, rmse = nfit, nrmse
        if done:
            break
    return T, fit, rmse, k


def transform_to_prediction(T):
    """train.py:473-481: translation = T[:3, 3]; angle = euler 'xyz' z component (= atan2(R10, R00) for a z rotation)."""
    return T[:3, 3].copy(), float(np.arctan2(T[1, 0], T[0, 0]))
