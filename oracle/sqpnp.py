"""TEST INFRASTRUCTURE ONLY -- fp64 restatement of SQPnP, the solver behind the reference's `cv2.solvePnPRansac(..., flags=cv2.SOLVEPNP_SQPNP)`
(fast3r/dust3r/cloud_opt/init_im_poses.py:335).  parity of the implementation: unpinned (OpenCV, an un-vendored and un-pinned dependency
of the reference, is not installable here); parity of the RESULT: SQPnP returns the global minimiser over SO(3) of a stated cost, so any
correct implementation returns the same pose on a well-posed problem -- that is what tests/test_pnp.py::test_sqpnp_* check (exact recovery
of known cameras, optimality against random rotations and against a dense local search).

Algorithm (G. Terzakis, M. Lourakis, "A Consistently Fast and Globally Optimal Solution to the Perspective-n-Point Problem", ECCV 2020;
OpenCV modules/calib3d/src/sqpnp.cpp is a port of the authors' code):
  * for points M_i with normalised projections (x_i, y_i):  cost(R, t) = sum_i (R M_i + t)^T Q_i (R M_i + t),
    Q_i = [[1, 0, -x_i], [0, 1, -y_i], [-x_i, -y_i, x_i^2 + y_i^2]]  (the squared distance of the transformed point to its viewing ray, scaled);
  * t is eliminated in closed form, t = P r with r = vec(R) (row-major) and P = -(sum Q_i)^-1 sum Q_i A_i, A_i r = R M_i,
    leaving cost = r^T Omega r, Omega = sum A_i^T Q_i A_i + (sum Q_i A_i)^T P  (9 x 9, positive semi-definite);
  * r^T Omega r is minimised over the rotation manifold by sequential quadratic programming started from +/- sqrt(3) x the eigenvectors of
    Omega with the smallest eigenvalues (projected to the nearest rotation); further eigenvectors are tried while the best cost found
    exceeds 3 x their eigenvalue (the bound that makes the search globally optimal); each SQP step solves the equality-constrained
    quadratic programme  min (r + d)^T Omega (r + d)  s.t.  J(r) d = -h(r)  for the six orthonormality constraints h;
  * candidates must put the point cloud in front of the camera; the result is the feasible candidate of least cost.
"""
import numpy as np

RANK_TOL = 1e-7            # eigenvalues of Omega below this are its null space (authors' default)
SQP_TOL = 1e-10            # squared step norm that ends the SQP iterations
SQP_MAX_ITER = 15
ORTHO_SQ_TOL = 1e-8        # an eigenvector this close to a (scaled) rotation is taken as is
EQUAL_SQ_ERR = 1e-10


def nearest_rotation(e):
    """the rotation closest (Frobenius) to the 3 x 3 matrix with row-major entries e"""
    U, _, Vt = np.linalg.svd(np.asarray(e, dtype=np.float64).reshape(3, 3))
    D = np.diag([1.0, 1.0, np.linalg.det(U @ Vt)])
    return (U @ D @ Vt).reshape(9)


def _constraints(r):
    """h(r) (6,) and its Jacobian J (6, 9) for the rows r1, r2, r3 of R: unit norms, mutual orthogonality"""
    r1, r2, r3 = r[0:3], r[3:6], r[6:9]
    h = np.array([r1 @ r1 - 1, r2 @ r2 - 1, r3 @ r3 - 1, r1 @ r2, r1 @ r3, r2 @ r3])
    J = np.zeros((6, 9))
    J[0, 0:3], J[1, 3:6], J[2, 6:9] = 2 * r1, 2 * r2, 2 * r3
    J[3, 0:3], J[3, 3:6] = r2, r1
    J[4, 0:3], J[4, 6:9] = r3, r1
    J[5, 3:6], J[5, 6:9] = r3, r2
    return h, J


def _sqp_step(Omega, r):
    """d minimising (r + d)^T Omega (r + d) subject to the linearised constraints J d = -h: d = x + N y with x the minimum-norm solution of
    the constraints (their row space) and y the minimiser of the quadratic over the null space N of J"""
    h, J = _constraints(r)
    x = np.linalg.lstsq(J, -h, rcond=None)[0]
    _, _, Vt = np.linalg.svd(J)
    N = Vt[6:].T                                     # (9, 3)
    A = N.T @ Omega @ N
    b = -N.T @ Omega @ (r + x)
    y = np.linalg.lstsq(A, b, rcond=None)[0]
    return x + N @ y


def run_sqp(Omega, r0):
    r = np.array(r0, dtype=np.float64)
    for _ in range(SQP_MAX_ITER):
        d = _sqp_step(Omega, r)
        r = r + d
        if d @ d < SQP_TOL:
            break
    det = np.linalg.det(r.reshape(3, 3))
    if det < 0:
        r, det = -r, -det
    return nearest_rotation(r)  # (the authors project only when det > 1.001; projecting always changes nothing at a converged point)


def omega_matrix(M, xy, w=None):
    """-> (Omega (9, 9), P (3, 9), mean point (3,)); None when sum Q_i is singular (all rays parallel)"""
    M = np.asarray(M, dtype=np.float64).reshape(-1, 3)
    xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
    n = M.shape[0]
    w = np.ones(n) if w is None else np.asarray(w, dtype=np.float64)
    x, y = xy[:, 0], xy[:, 1]
    Q = np.zeros((n, 3, 3))
    Q[:, 0, 0] = Q[:, 1, 1] = 1.0
    Q[:, 0, 2] = Q[:, 2, 0] = -x
    Q[:, 1, 2] = Q[:, 2, 1] = -y
    Q[:, 2, 2] = x * x + y * y
    Q *= w[:, None, None]
    A = np.zeros((n, 3, 9))
    A[:, 0, 0:3] = A[:, 1, 3:6] = A[:, 2, 6:9] = M
    QA = np.einsum("nij,njk->ik", Q, A)
    Qsum = Q.sum(0)
    if abs(np.linalg.det(Qsum)) < 1e-300:
        return None
    P = -np.linalg.solve(Qsum, QA)
    Omega = np.einsum("nji,njk,nkl->il", A, Q, A) + QA.T @ P
    Omega = 0.5 * (Omega + Omega.T)
    return Omega, P, (M * w[:, None]).sum(0) / w.sum()


def solve(M, xy):
    """SQPnP: world points M (n, 3), normalised image points xy (n, 2), n >= 3 -> (R (3, 3), t (3,), squared error) world-to-camera,
    or None when the problem is degenerate / no candidate puts the points in front of the camera."""
    om = omega_matrix(M, xy)
    if om is None:
        return None
    Omega, P, mean = om
    s, U = np.linalg.eigh(Omega)                      # ascending eigenvalues
    num_null = int((s < RANK_TOL).sum())
    if num_null > 6:
        return None
    best = [None, np.inf]

    def handle(r_hat):
        t = P @ r_hat
        if r_hat[6:9] @ mean + t[2] <= 0:             # cheirality on the centroid (authors' test)
            return
        err = float(r_hat @ Omega @ r_hat)
        if err < best[1] - EQUAL_SQ_ERR or best[0] is None:
            best[0], best[1] = (r_hat.copy(), t.copy()), err

    def try_vector(e):
        e = np.sqrt(3.0) * e
        h, _ = _constraints(e)
        if h @ h < ORTHO_SQ_TOL:
            handle(e * np.sign(np.linalg.det(e.reshape(3, 3))))
            return
        for sgn in (1.0, -1.0):
            handle(run_sqp(Omega, nearest_rotation(sgn * e)))

    num_eigen = num_null if num_null > 0 else 1
    for i in range(num_eigen):
        try_vector(U[:, i])
    idx = num_eigen
    while idx < 9 and best[1] > 3.0 * s[idx]:
        try_vector(U[:, idx])
        idx += 1
    if best[0] is None:
        return None
    r_hat, t = best[0]
    return r_hat.reshape(3, 3), t, best[1]
