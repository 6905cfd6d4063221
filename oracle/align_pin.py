"""TEST INFRASTRUCTURE ONLY -- an INDEPENDENT closed form for the similarity registration that `align_local_pts3d_to_global` solves
(fast3r/models/multiview_dust3r_module.py:509-511: roma.rigid_points_registration(x, y, compute_scaling=True)).

The reference's solver lives in the un-vendored, un-pinned `roma` package (not installable here: no network), so the restatement in
oracle/align_oracle.py (Umeyama 1991: SVD of the cross-covariance with the det correction) cannot be run against the reference itself.
It is pinned instead against a DIFFERENT derivation of the same least-squares problem

    min over proper rotations R, s > 0, t of  sum_k | y_k - (s R x_k + t) |^2

namely Horn's unit-quaternion solution (B. K. P. Horn, "Closed-form solution of absolute orientation using unit quaternions", JOSA A
4(4), 1987): the optimal rotation is the eigenvector of the largest eigenvalue of a symmetric 4 x 4 matrix N built from the
cross-covariance (section 4.A, eq. for N), the scale is the asymmetric least-squares one sum y'.(R x') / sum |x'|^2 (section 2.D),
the translation ym - s R xm (section 2.C).  No SVD, no determinant fix: a quaternion is a proper rotation by construction, which is
exactly the constraint roma's `special_procrustes` enforces with its sign flip.  Everything in float64.

Two algorithms that share no code and agree to 1e-10 on generic, mirrored, planar and noisy clouds pin each other; the HIP path
(f3r_post.hip: raw fp64 moments + Jacobi SVD) is then checked against both (tests/test_align.py).
"""
import torch


def horn_similarity(x: torch.Tensor, y: torch.Tensor):
    """y ~ s R x + t, x, y (M, 3) -> (R (3,3), t (3,), s) in float64."""
    x, y = x.double(), y.double()
    xm, ym = x.mean(0), y.mean(0)
    xc, yc = x - xm, y - ym
    S = xc.t() @ yc  # S[a][b] = sum_k xc[k][a] * yc[k][b]   (Horn's M)
    Sxx, Sxy, Sxz = S[0, 0], S[0, 1], S[0, 2]
    Syx, Syy, Syz = S[1, 0], S[1, 1], S[1, 2]
    Szx, Szy, Szz = S[2, 0], S[2, 1], S[2, 2]
    N = torch.stack([
        torch.stack([Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx]),
        torch.stack([Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz]),
        torch.stack([Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy]),
        torch.stack([Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz]),
    ])
    w, v = torch.linalg.eigh(N)
    q0, qx, qy, qz = v[:, -1]  # unit quaternion of the largest eigenvalue
    R = torch.stack([
        torch.stack([q0 * q0 + qx * qx - qy * qy - qz * qz, 2 * (qx * qy - q0 * qz), 2 * (qx * qz + q0 * qy)]),
        torch.stack([2 * (qy * qx + q0 * qz), q0 * q0 - qx * qx + qy * qy - qz * qz, 2 * (qy * qz - q0 * qx)]),
        torch.stack([2 * (qz * qx - q0 * qy), 2 * (qz * qy + q0 * qx), q0 * q0 - qx * qx - qy * qy + qz * qz]),
    ])
    s = (yc * (xc @ R.t())).sum() / (xc ** 2).sum()
    t = ym - s * (R @ xm)
    return R, t, s
