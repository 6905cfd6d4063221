"""TEST INFRASTRUCTURE ONLY -- a stand-in for the three OpenCV names the reference's `fast_pnp` touches
(fast3r/dust3r/cloud_opt/init_im_poses.py:300-350): `cv2.solvePnPRansac(..., flags=cv2.SOLVEPNP_SQPNP)`, `cv2.Rodrigues`, `cv2.error`.

OpenCV is not installable here, so the camera-pose row cannot be compared against the reference's own
solver.  What CAN run for real is everything the reference wraps around it: `MultiViewDUSt3RLitModule.estimate_camera_poses`
(multiview_dust3r_module.py:807-869), `estimate_cam_pose_one_sample` (:1038-1078: the `conf > 1.0` mask, identity pose on failure), the
reference's `estimate_focal`, and `fast_pnp` itself (pixel grid, principal point, the np.geomspace focal candidates and their selection by
inlier count, Rodrigues -> 4x4 -> inverse).  oracle/make_golden_pose.py imports those with this module installed as `cv2` and stores
their results as a fixture; the HIP path is compared with it in tests/test_pnp.py.

The solver below is deliberately NOT the product's algorithm (fast3r_amd/csrc/f3r_pnp.hip: 32 fixed 6-point samples, one 4x4
eigenproblem shared by all focals, gated Gauss-Newton): numpy, RANSAC over random 6-point samples with a calibrated DLT on Hartley-
normalised points, refit on the inliers, Gauss-Newton in the rotation-vector parametrisation.  Same contract as OpenCV's call:
(success, rvec (3,1), tvec (3,1), inliers (n,1) int32) with world-to-camera [R | t]."""
import numpy as np

SOLVEPNP_SQPNP = 8


def __getattr__(name):  # anything else the reference merely mentions at import time (cv2.IMREAD_COLOR in a default argument, ...)
    if name.startswith("__"):
        raise AttributeError(name)
    return 0


class error(Exception):
    pass


def Rodrigues(src):
    """rotation vector (3,) / (3,1) -> (R (3,3) float64, None);  R (3,3) -> (rvec (3,1), None)"""
    a = np.asarray(src, dtype=np.float64)
    if a.shape == (3, 3):
        cos = np.clip((np.trace(a) - 1.0) / 2.0, -1.0, 1.0)
        th = np.arccos(cos)
        w = np.array([a[2, 1] - a[1, 2], a[0, 2] - a[2, 0], a[1, 0] - a[0, 1]])
        if th < 1e-12:
            return (0.5 * w).reshape(3, 1), None
        if np.pi - th < 1e-6:  # near pi: axis from the symmetric part
            B = (a + np.eye(3)) / 2.0
            ax = np.sqrt(np.clip(np.diag(B), 0.0, None))
            i = int(np.argmax(ax))
            ax = B[:, i] / ax[i]
            return (th * ax / np.linalg.norm(ax)).reshape(3, 1), None
        return (th / (2.0 * np.sin(th)) * w).reshape(3, 1), None
    r = a.reshape(3)
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3), None
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx), None


def _dlt(X, xn):
    """calibrated DLT: X (n,3) world points, xn (n,2) normalised image coordinates -> (R, t) world-to-camera, or None"""
    n = X.shape[0]
    c = X.mean(0)
    s = np.sqrt(((X - c) ** 2).sum(1).mean()) + 1e-30
    Xh = np.concatenate([(X - c) / s, np.ones((n, 1))], 1)
    A = np.zeros((2 * n, 12))
    A[0::2, 0:4] = Xh
    A[0::2, 8:12] = -xn[:, 0:1] * Xh
    A[1::2, 4:8] = Xh
    A[1::2, 8:12] = -xn[:, 1:2] * Xh
    try:
        _, _, Vt = np.linalg.svd(A, full_matrices=False)
    except np.linalg.LinAlgError:
        return None
    P = Vt[-1].reshape(3, 4)
    if (Xh @ P[2]).sum() < 0:  # points in front of the camera
        P = -P
    U, S, Wt = np.linalg.svd(P[:, :3])
    R = U @ Wt
    if np.linalg.det(R) < 0:
        return None
    sc = S.mean()
    if sc < 1e-12:
        return None
    t_n = P[:, 3] / sc  # x ~ R (X - c) / s + t_n  ->  scale the whole equation by s
    t = s * t_n - R @ c
    return R, t


def _project(X, R, t, K):
    Xc = X @ R.T + t
    z = Xc[:, 2:3]
    uv = Xc[:, :2] / np.where(np.abs(z) < 1e-12, 1e-12, z) * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]])
    return uv, Xc


def _refine(X, uv, R, t, K, iters=8):
    f = np.array([K[0, 0], K[1, 1]])
    for _ in range(iters):
        p, Xc = _project(X, R, t, K)
        r = (p - uv).reshape(-1)
        z = Xc[:, 2]
        RX = Xc - t
        J = np.zeros((X.shape[0], 2, 6))
        du = np.stack([f[0] / z, np.zeros_like(z), -f[0] * Xc[:, 0] / z ** 2], 1)
        dv = np.stack([np.zeros_like(z), f[1] / z, -f[1] * Xc[:, 1] / z ** 2], 1)
        # d Xc / d delta (left perturbation exp(delta^) R) = -[R X]_x ;  d Xc / d t = I
        skew = np.zeros((X.shape[0], 3, 3))
        skew[:, 0, 1], skew[:, 0, 2] = -RX[:, 2], RX[:, 1]
        skew[:, 1, 0], skew[:, 1, 2] = RX[:, 2], -RX[:, 0]
        skew[:, 2, 0], skew[:, 2, 1] = -RX[:, 1], RX[:, 0]
        J[:, 0, :3] = -np.einsum("ni,nij->nj", du, skew)
        J[:, 1, :3] = -np.einsum("ni,nij->nj", dv, skew)
        J[:, 0, 3:], J[:, 1, 3:] = du, dv
        Jm = J.reshape(-1, 6)
        H = Jm.T @ Jm + 1e-9 * np.eye(6)
        try:
            d = np.linalg.solve(H, -Jm.T @ r)
        except np.linalg.LinAlgError:
            break
        R = Rodrigues(d[:3])[0] @ R
        t = t + d[3:]
        if np.abs(d).max() < 1e-12:
            break
    return R, t


def solvePnPRansac(objectPoints, imagePoints, cameraMatrix, distCoeffs, iterationsCount=100, reprojectionError=8.0, flags=0, **_):
    X = np.asarray(objectPoints, dtype=np.float64).reshape(-1, 3)
    uv = np.asarray(imagePoints, dtype=np.float64).reshape(-1, 2)
    K = np.asarray(cameraMatrix, dtype=np.float64)
    n = X.shape[0]
    if n < 4 or uv.shape[0] != n:
        raise error("solvePnPRansac: need >= 4 correspondences")
    if n < 6:
        return False, None, None, None
    xn = (uv - np.array([K[0, 2], K[1, 2]])) / np.array([K[0, 0], K[1, 1]])
    rng = np.random.default_rng(20240917)
    best = None
    for _ in range(max(int(iterationsCount), 1)):
        idx = rng.choice(n, 6, replace=False)
        sol = _dlt(X[idx], xn[idx])
        if sol is None:
            continue
        p, Xc = _project(X, sol[0], sol[1], K)
        inl = (np.linalg.norm(p - uv, axis=1) < reprojectionError) & (Xc[:, 2] > 0)
        if best is None or inl.sum() > best[0].sum():
            best = (inl, sol)
    if best is None or best[0].sum() < 6:
        return False, None, None, None
    inl, (R, t) = best
    for _ in range(2):  # refit on the consensus set, refine, re-select
        sol = _dlt(X[inl], xn[inl])
        if sol is not None:
            R, t = sol
        R, t = _refine(X[inl], uv[inl], R, t, K)
        p, Xc = _project(X, R, t, K)
        new = (np.linalg.norm(p - uv, axis=1) < reprojectionError) & (Xc[:, 2] > 0)
        if new.sum() < 6:
            break
        inl = new
    return True, Rodrigues(R)[0], t.reshape(3, 1), np.flatnonzero(inl).astype(np.int32).reshape(-1, 1)
