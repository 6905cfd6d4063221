"""TEST INFRASTRUCTURE ONLY -- a stand-in for the three OpenCV names the reference's `fast_pnp` touches
(fast3r/dust3r/cloud_opt/init_im_poses.py:300-350): `cv2.solvePnPRansac(..., flags=cv2.SOLVEPNP_SQPNP)`, `cv2.Rodrigues`, `cv2.error`.

OpenCV is not installable here (an un-vendored, un-pinned dependency of the reference).  What runs for real is everything the reference
wraps around it: `MultiViewDUSt3RLitModule.estimate_camera_poses` (multiview_dust3r_module.py:807-869), `estimate_cam_pose_one_sample`
(:1038-1078: the `conf > 1.0` mask, identity pose on failure), the reference's `estimate_focal`, and `fast_pnp` itself (pixel grid,
principal point, the np.geomspace focal candidates and their selection by inlier count, Rodrigues -> 4x4 -> inverse).
oracle/make_golden_pose.py imports those with this module installed as `cv2` and stores their results as a fixture; the HIP path is
compared with it in tests/test_pnp.py.

The solver restates the NAMED dependency's published algorithm, not the product's (fast3r_amd/csrc/f3r_pnp.hip: sampled 6-point DLT
hypotheses + gated Gauss-Newton on the reprojection error):
  * `solvePnPRansac` follows OpenCV's structure (modules/calib3d/src/solvepnp.cpp): `iterationsCount` random 5-point samples, one model per
    sample, inliers = squared reprojection error <= reprojectionError^2, best model by inlier count, adaptive iteration count at
    confidence 0.99 (RANSACUpdateNumIters), then the FINAL pose = the flagged solver -- SQPnP -- on all inliers of the best model; the
    returned inlier list is the best model's.  Two stated differences: the 5-point models come from SQPnP too (OpenCV uses EPnP there: it
    only has to find the consensus set), and the sampler is numpy's generator with a fixed seed (OpenCV's RNG is not reproducible here);
  * SQPnP itself is oracle/sqpnp.py (fp64; Terzakis & Lourakis, ECCV 2020): the global minimiser of its cost over SO(3), so a correct
    implementation is pinned by the problem, not by implementation details.
Same contract as OpenCV's call: (success, rvec (3,1), tvec (3,1), inliers (n,1) int32) with world-to-camera [R | t]."""
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


def _project(X, R, t, K):
    Xc = X @ R.T + t
    z = Xc[:, 2:3]
    uv = Xc[:, :2] / np.where(np.abs(z) < 1e-12, 1e-12, z) * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]])
    return uv, Xc


def _update_num_iters(p, ep, model_points, max_iters):
    """cv::RANSACUpdateNumIters: iterations needed to draw one outlier-free sample with confidence p at outlier ratio ep"""
    p = min(max(p, 0.0), 1.0)
    ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, np.finfo(np.float64).tiny)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < np.finfo(np.float64).tiny:
        return 0
    num, denom = np.log(num), np.log(denom)
    return max_iters if (denom >= 0 or -num >= max_iters * (-denom)) else int(round(num / denom))


def solvePnPRansac(objectPoints, imagePoints, cameraMatrix, distCoeffs, iterationsCount=100, reprojectionError=8.0, confidence=0.99,
                   flags=0, **_):
    from oracle import sqpnp
    X = np.asarray(objectPoints, dtype=np.float64).reshape(-1, 3)
    uv = np.asarray(imagePoints, dtype=np.float64).reshape(-1, 2)
    K = np.asarray(cameraMatrix, dtype=np.float64)
    n = X.shape[0]
    if n < 4 or uv.shape[0] != n:
        raise error("solvePnPRansac: need >= 4 correspondences")
    xn = (uv - np.array([K[0, 2], K[1, 2]])) / np.array([K[0, 0], K[1, 1]])  # undistortPoints with no distortion
    model_points = 5 if n > 4 else 4
    thr2 = float(reprojectionError) ** 2
    rng = np.random.default_rng(20240917)
    niters = max(int(iterationsCount), 1)
    best_mask, best_count = None, 0
    it = 0
    while it < niters:
        it += 1
        idx = rng.choice(n, model_points, replace=False)
        sol = sqpnp.solve(X[idx], xn[idx])
        if sol is None:
            continue
        p, Xc = _project(X, sol[0], sol[1], K)
        mask = ((p - uv) ** 2).sum(1) <= thr2
        cnt = int(mask.sum())
        if cnt > max(best_count, model_points - 1):
            best_mask, best_count = mask, cnt
            niters = _update_num_iters(confidence, (n - cnt) / n, model_points, niters)
    if best_mask is None:
        return False, None, None, None
    sol = sqpnp.solve(X[best_mask], xn[best_mask])  # the flagged solver on the consensus set
    if sol is None:
        return False, None, None, None
    R, t, _ = sol
    return True, Rodrigues(R)[0], t.reshape(3, 1), np.flatnonzero(best_mask).astype(np.int32).reshape(-1, 1)
