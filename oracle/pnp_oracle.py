"""TEST INFRASTRUCTURE ONLY -- CPU (torch, fp64) restatement of the pose step of `estimate_camera_poses`
(fast3r/models/multiview_dust3r_module.py:807-869 -> estimate_cam_pose_one_sample :1038-1078 -> fast_pnp,
fast3r/dust3r/cloud_opt/init_im_poses.py:300-350).

The reference's control flow is restated line by line (mask = conf > 1.0 :1045, focal search over np.geomspace(S/2, 3S, 100) when no
focal is given :312-316, best candidate by inlier count at 5 px :330-343, cam-to-world = inverse of the world-to-cam [R|T] :349-350,
identity pose on failure :1062-1064).  Its arithmetic, however, is `cv2.solvePnPRansac(..., flags=cv2.SOLVEPNP_SQPNP)` -- OpenCV is
neither vendored nor installed here and its RANSAC draws from its own RNG -- so the solver below is NOT a restatement of OpenCV: it is
the algorithm of the HIP path (f3r_post.hip::pnp_*), written independently on torch.linalg:
    1. calibrated DLT in closed form for every focal at once: with P = [X 1] and centred pixels (px, py) the normal matrix of
       {r1.P - (px/f) r3.P = 0, r2.P - (py/f) r3.P = 0} is [[S0,0,-S1x/f],[0,S0,-S1y/f],[.,.,S2/f^2]]; eliminating the first two
       blocks leaves (S2 - S1x S0^-1 S1x - S1y S0^-1 S1y) c = lambda c for the third row c = [r3 t3] INDEPENDENT of f, and
       [r1 t1] = S0^-1 S1x c / f, [r2 t2] = S0^-1 S1y c / f;
    2. nearest rotation (SVD) + scale, positive-depth sign;  3. score = points within 5 px;  4. gated Gauss-Newton on the best;
    5. (round 3) SQPnP on the points within 5 px of that pose: the final solve of cv2.solvePnPRansac(SOLVEPNP_SQPNP), restated in oracle/sqpnp.py.
PARITY UNPINNED against the reference for this row; anchored on ground truth instead (tests/test_pnp.py: known poses are recovered) and
on HIP == this file.
"""
import math

import numpy as np
import torch

from oracle import sqpnp

REPROJ_THR = 5.0  # init_im_poses.py:335
N_GN = 6


def _moments(P, px, py):
    S0 = P.t() @ P
    S1x = P.t() @ (P * px[:, None])
    S1y = P.t() @ (P * py[:, None])
    S2 = P.t() @ (P * (px * px + py * py)[:, None])
    return S0, S1x, S1y, S2


def _pose_from_rows(a, b, c, f):
    """rows of [R|t] up to scale -> proper rotation, translation (world -> camera)."""
    A = torch.stack([a[:3] / f, b[:3] / f, c[:3]])
    t = torch.stack([a[3] / f, b[3] / f, c[3]])
    U, S, Vh = torch.linalg.svd(A)
    d = torch.sign(torch.linalg.det(U @ Vh))
    D = torch.diag(torch.tensor([1.0, 1.0, float(d)], dtype=A.dtype))
    R = U @ D @ Vh
    s = (S[0] + S[1] + float(d) * S[2]) / 3.0
    return R, t / s


def _project(R, t, X, f):
    Xc = X @ R.t() + t
    z = Xc[:, 2]
    return f * Xc[:, 0] / z, f * Xc[:, 1] / z, z


def _count_inliers(R, t, X, px, py, f):
    """(inliers at 5 px, truncated squared error): the count is the reference's score (:342); on clean scenes it saturates over a
    range of focals (focal / depth ambiguity), so ties are broken by the MSAC cost instead of by candidate order."""
    u, v, z = _project(R, t, X, f)
    e2 = (u - px) ** 2 + (v - py) ** 2
    inl = (e2 <= REPROJ_THR ** 2) & (z > 0)
    cost = float(torch.where(inl, e2, torch.full_like(e2, REPROJ_THR ** 2)).sum())
    return int(inl.sum()), cost


def _skew(w):
    return torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=w.dtype)


def _expm_so3(w):
    th = float(w.norm())
    K = _skew(w)
    if th < 1e-12:
        return torch.eye(3, dtype=w.dtype) + K
    return torch.eye(3, dtype=w.dtype) + math.sin(th) / th * K + (1 - math.cos(th)) / th ** 2 * (K @ K)


def _refine(R, t, X, px, py, f):
    for _ in range(N_GN):
        Xc = X @ R.t() + t
        x, y, z = Xc[:, 0], Xc[:, 1], Xc[:, 2]
        ex, ey = f * x / z - px, f * y / z - py
        m = ((ex * ex + ey * ey) <= REPROJ_THR ** 2) & (z > 0)
        if int(m.sum()) < 4:
            break
        x, y, z, ex, ey = x[m], y[m], z[m], ex[m], ey[m]
        iz = 1.0 / z
        # d(proj)/d(Xc): [[f/z, 0, -f x/z^2], [0, f/z, -f y/z^2]];  d(Xc)/d(w) = -[Xc]_x,  d(Xc)/d(t) = I
        Jx = torch.stack([-f * x * y * iz * iz, f * (1 + x * x * iz * iz), -f * y * iz, f * iz, torch.zeros_like(z), -f * x * iz * iz], 1)
        Jy = torch.stack([-f * (1 + y * y * iz * iz), f * x * y * iz * iz, f * x * iz, torch.zeros_like(z), f * iz, -f * y * iz * iz], 1)
        Hm = Jx.t() @ Jx + Jy.t() @ Jy
        g = Jx.t() @ ex + Jy.t() @ ey
        Hm = Hm + 1e-9 * torch.diag(torch.diag(Hm)) + 1e-12 * torch.eye(6, dtype=Hm.dtype)
        d = -torch.linalg.solve(Hm, g)
        dR = _expm_so3(d[:3])
        R = dR @ R
        t = dR @ t + d[3:]
    return R, t


N_HYP = 32      # deterministic 6-point samples
SAMPLE = 6


def sample_index(h, j, npix):
    """pseudo-random pixel of sample h, slot j (same integer recipe in f3r_post.hip)."""
    x = (1103515245 * (h * SAMPLE + j + 1) + 12345) & 0x7FFFFFFF
    x = (x * 2654435761) & 0xFFFFFFFF
    return x % npix


def _dlt_rows(S0, S1x, S1y, S2):
    """closed-form calibrated DLT (see the module docstring): rows (a, b, c) with a, b still to be divided by the focal."""
    S0i = torch.linalg.inv(S0)
    G = S2 - S1x @ S0i @ S1x - S1y @ S0i @ S1y
    G = 0.5 * (G + G.t())
    evals, evecs = torch.linalg.eigh(G)
    c = evecs[:, 0]
    a1, b1 = S0i @ S1x @ c, S0i @ S1y @ c
    if float(c @ S0[:, 3]) < 0:  # sum of the depths of the points that built the moments must be positive
        a1, b1, c = -a1, -b1, -c
    return a1, b1, c


def fast_pnp(pts3d, focal, msk, pp=None, num_guessed_focals=100):
    """init_im_poses.py:300-350.  pts3d (H, W, 3), msk (H, W) bool -> (best_focal, cam_to_world 4x4 fp64 tensor) or (None, None)."""
    if int(msk.sum()) < 4:  # :302-303
        return None, None
    H, W, _ = pts3d.shape
    npix = H * W
    if pp is None:
        pp = (W / 2, H / 2)  # :318-319
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    mflat = msk.reshape(-1)
    Xall = pts3d.double().reshape(-1, 3)
    pxall = xs.reshape(-1) - float(pp[0])
    pyall = ys.reshape(-1) - float(pp[1])
    X, px, py = Xall[mflat], pxall[mflat], pyall[mflat]
    if focal is None:  # :312-316
        S = max(W, H)
        cands = [float(v) for v in np.geomspace(S / 2, S * 3, num=num_guessed_focals)]
    else:
        cands = [float(focal)]
    # conditioning: centre / scale the world points (undone in `denorm`)
    cen = X.mean(0)
    sig = float((X - cen).square().sum(1).mean().sqrt().clamp_min(1e-12))

    def homog(Xs):
        return torch.cat([(Xs - cen) / sig, torch.ones((Xs.shape[0], 1), dtype=torch.float64)], 1)

    def denorm(Rn, tn):
        return Rn, sig * tn - Rn @ cen  # Xc ~ Rn (X - cen) / sig + tn, common scale dropped

    def nearest_cand(f):
        return min(cands, key=lambda v: abs(math.log(v) - math.log(max(f, 1e-9))))

    # ---- stage 1: N_HYP deterministic 6-point hypotheses, each scored on ALL points (the role RANSAC plays in the reference)
    best = (0, math.inf, None, None, None)
    if int(mflat.sum()) >= SAMPLE:
        for h in range(N_HYP):
            idxs = []
            for j in range(SAMPLE):
                i = sample_index(h, j, npix)
                while not bool(mflat[i]):
                    i = (i + 1) % npix
                idxs.append(i)
            idx = torch.tensor(idxs)
            S0, S1x, S1y, S2 = _moments(homog(Xall[idx]), pxall[idx], pyall[idx])
            if float(torch.linalg.det(S0)) < 1e-12:
                continue
            a1, b1, c = _dlt_rows(S0, S1x, S1y, S2)
            nc = float(c[:3].norm())
            if nc < 1e-12:
                continue
            f_h = cands[0] if len(cands) == 1 else nearest_cand(float(a1[:3].norm()) / nc)
            R, t = denorm(*_pose_from_rows(a1, b1, c, f_h))
            score, cost = _count_inliers(R, t, X, px, py, f_h)
            if score > best[0] or (score == best[0] and cost < best[1]):
                best = (score, cost, R, t, f_h)
    # ---- stage 2: DLT on the inliers of the best hypothesis (all points when there is none), every candidate focal scored
    if best[0] >= SAMPLE:
        _, _, R, t, f_h = best
        u, v, z = _project(R, t, X, f_h)
        inl = (((u - px) ** 2 + (v - py) ** 2) <= REPROJ_THR ** 2) & (z > 0)
    else:
        inl = torch.ones(X.shape[0], dtype=torch.bool)
    a1, b1, c = _dlt_rows(*_moments(homog(X[inl]), px[inl], py[inl]))
    if len(cands) == 1:
        f = cands[0]
    else:
        # the DLT is uncalibrated in disguise: rows 1, 2 come out multiplied by the focal, so |a| / |c| and |b| / |c| estimate it
        # directly.  Take the grid candidate nearest to their geometric mean, then let the inlier count (the reference's score, :342)
        # and the truncated cost decide among it and its two neighbours.
        nc = float(c[:3].norm())
        f_dlt = math.sqrt(max(float(a1[:3].norm()) * float(b1[:3].norm()), 1e-30)) / max(nc, 1e-30)
        k0 = cands.index(nearest_cand(f_dlt))
        best = (0, math.inf, None)
        for k in range(max(0, k0 - 1), min(len(cands), k0 + 2)):
            R, t = denorm(*_pose_from_rows(a1, b1, c, cands[k]))
            R, t = _refine(R, t, X, px, py, cands[k])
            score, cost = _count_inliers(R, t, X, px, py, cands[k])
            if score > best[0] or (score == best[0] and cost < best[1]):
                best = (score, cost, cands[k])
        if not best[0]:
            return None, None
        f = best[2]
    R, t = denorm(*_pose_from_rows(a1, b1, c, f))
    if _count_inliers(R, t, X, px, py, f)[0] == 0:
        return None, None
    R, t = _refine(R, t, X, px, py, f)
    # ---- stage 3: SQPnP (oracle/sqpnp.py) on the consensus set of that pose -- what cv2.solvePnPRansac(SOLVEPNP_SQPNP) ends with
    u, v, z = _project(R, t, X, f)
    inl = (((u - px) ** 2 + (v - py) ** 2) <= REPROJ_THR ** 2) & (z > 0)
    sol = sqpnp.solve(X[inl].numpy(), torch.stack([px[inl] / f, py[inl] / f], 1).numpy()) if int(inl.sum()) >= 3 else None
    if sol is not None:
        R, t = torch.from_numpy(sol[0]), torch.from_numpy(sol[1])
    T = torch.eye(4, dtype=torch.float64)
    T[:3, :3] = R.t()
    T[:3, 3] = -R.t() @ t
    return f, T


def estimate_cam_pose_one_sample(sample_preds, focal_key="focal_length"):
    """multiview_dust3r_module.py:1038-1078."""
    poses, focals = [], []
    for pred in sample_preds:
        pts3d = pred["pts3d_in_other_view"].squeeze(0) if pred["pts3d_in_other_view"].dim() == 4 else pred["pts3d_in_other_view"]
        conf = pred["conf"].squeeze(0) if pred["conf"].dim() == 3 else pred["conf"]
        msk = conf > 1.0  # :1045
        f0 = float(pred[focal_key]) if focal_key in pred else None  # :1049
        f, T = fast_pnp(pts3d, f0, msk)
        if T is None or f is None:  # :1062-1064
            poses.append(np.eye(4))
            focals.append(f)
        else:
            poses.append(T.numpy())
            focals.append(f)
    return poses, focals
