// SQPnP (G. Terzakis, M. Lourakis, "A Consistently Fast and Globally Optimal Solution to the Perspective-n-Point Problem", ECCV 2020) in
// fp64, one thread per problem: the solver cv2.solvePnPRansac(..., flags=cv2.SOLVEPNP_SQPNP) runs on its consensus set
// (fast3r/dust3r/cloud_opt/init_im_poses.py:335; OpenCV's solvePnPRansac ends with solvePnP(inliers, flags)), so the pose the reference
// returns is the minimiser over SO(3) of
//     cost(R, t) = sum_i (R M_i + t)^T Q_i (R M_i + t),   Q_i = [[1, 0, -x_i], [0, 1, -y_i], [-x_i, -y_i, x_i^2 + y_i^2]]
// over the inliers (M_i world point, (x_i, y_i) its normalised image point).  f3r_pnp.hip accumulates the 40 sums below over the
// inliers of its own consensus set and calls solve(); the same function compiles for the host (tests/test_pnp.py builds it with g++
// and checks it against the fp64 restatement oracle/sqpnp.py on the same sums).
//
// With r = vec(R) row-major, A_i r = R M_i, the translation eliminated in closed form (t = P r, P = -(sum Q)^-1 sum Q A) the cost is
// r^T Omega r, Omega = sum A^T Q A + (sum Q A)^T P: a 9 x 9 positive semi-definite matrix whose blocks are sums of w * M M^T and w * M
// for w in {1, x, y, x^2 + y^2}.  It is minimised by sequential quadratic programming on the six orthonormality constraints, started
// from +/- sqrt(3) x the eigenvectors of Omega with the smallest eigenvalues; further eigenvectors are tried while the best cost found
// exceeds 3 x their eigenvalue (the bound that makes the search global); candidates must put the centroid in front of the camera.
#pragma once
#include "f3r_linalg.h"

namespace f3r_sqpnp {

constexpr int N_SUMS = 40;            // for w in {1, x, y, q = x^2 + y^2} (in this order): w, w M0, w M1, w M2, w M0M0, w M0M1, w M0M2, w M1M1, w M1M2, w M2M2
constexpr double RANK_TOL = 1e-7;     // eigenvalues of Omega below this are its null space (authors' default)
constexpr double SQP_TOL = 1e-10;     // squared step norm that ends the SQP iterations
constexpr int SQP_MAX_ITER = 15;
constexpr double ORTHO_SQ_TOL = 1e-8; // an eigenvector this close to a (scaled) rotation is taken as is
constexpr double EQUAL_SQ_ERR = 1e-10;

struct Result {
  double R[3][3];  // world -> camera
  double t[3];
  double err;      // r^T Omega r
  int ok;
};

F3R_LA_FN void accumulate(double* s, const double M[3], double x, double y) {
  const double w[4] = {1.0, x, y, x * x + y * y};
  const double mm[10] = {1.0, M[0], M[1], M[2], M[0] * M[0], M[0] * M[1], M[0] * M[2], M[1] * M[1], M[1] * M[2], M[2] * M[2]};
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 10; ++b) s[a * 10 + b] += w[a] * mm[b];
}

// the rotation closest (Frobenius) to the 3 x 3 matrix with row-major entries e
F3R_LA_FN void nearest_rotation(const double e[9], double r[9]) {
  double M[3][3], U[3][3], S[3], V[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[i][j] = e[i * 3 + j];
  f3r_la::svd3(M, U, S, V);
  double UVt[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) UVt[i][j] = U[i][0] * V[j][0] + U[i][1] * V[j][1] + U[i][2] * V[j][2];
  const double d = f3r_la::det3(UVt) < 0 ? -1.0 : 1.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r[i * 3 + j] = U[i][0] * V[j][0] + U[i][1] * V[j][1] + d * U[i][2] * V[j][2];
}

F3R_LA_FN double det9(const double r[9]) {
  return r[0] * (r[4] * r[8] - r[5] * r[7]) - r[1] * (r[3] * r[8] - r[5] * r[6]) + r[2] * (r[3] * r[7] - r[4] * r[6]);
}

// h(r) (6) and its Jacobian J (6 x 9) for the rows r1, r2, r3 of R: unit norms, mutual orthogonality
F3R_LA_FN void constraints(const double r[9], double h[6], double J[6][9]) {
  const double* r1 = r;
  const double* r2 = r + 3;
  const double* r3 = r + 6;
  auto dot = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
  h[0] = dot(r1, r1) - 1; h[1] = dot(r2, r2) - 1; h[2] = dot(r3, r3) - 1;
  h[3] = dot(r1, r2); h[4] = dot(r1, r3); h[5] = dot(r2, r3);
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 9; ++j) J[i][j] = 0.0;
  for (int k = 0; k < 3; ++k) {
    J[0][k] = 2 * r1[k]; J[1][3 + k] = 2 * r2[k]; J[2][6 + k] = 2 * r3[k];
    J[3][k] = r2[k]; J[3][3 + k] = r1[k];
    J[4][k] = r3[k]; J[4][6 + k] = r1[k];
    J[5][3 + k] = r3[k]; J[5][6 + k] = r2[k];
  }
}

// x with A x = b for a general 3 x 3 system; a (numerically) dependent direction gets 0 (the least-squares solution of least norm is
// not needed: at a non-degenerate point the reduced Hessian is positive definite)
F3R_LA_FN void solve3(double A[3][3], double b[3], double x[3]) {
  int perm[3] = {0, 1, 2};
  double scale = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) scale = fmax(scale, fabs(A[i][j]));
  bool dead[3] = {false, false, false};
  for (int c = 0; c < 3; ++c) {
    int p = c;
    for (int r = c + 1; r < 3; ++r)
      if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
    if (p != c) {
      for (int j = 0; j < 3; ++j) { const double tmp = A[c][j]; A[c][j] = A[p][j]; A[p][j] = tmp; }
      const double tb = b[c]; b[c] = b[p]; b[p] = tb;
    }
    if (fabs(A[c][c]) <= 1e-14 * scale || scale == 0.0) { dead[c] = true; continue; }
    for (int r = c + 1; r < 3; ++r) {
      const double f = A[r][c] / A[c][c];
      for (int j = c; j < 3; ++j) A[r][j] -= f * A[c][j];
      b[r] -= f * b[c];
    }
  }
  (void)perm;
  for (int c = 2; c >= 0; --c) {
    if (dead[c]) { x[c] = 0.0; continue; }
    double s = b[c];
    for (int j = c + 1; j < 3; ++j) s -= A[c][j] * x[j];
    x[c] = s / A[c][c];
  }
}

// d minimising (r + d)^T Omega (r + d) subject to the linearised constraints J d = -h: d = x + N y with x the minimum-norm solution of
// the constraints (their row space) and y the minimiser of the quadratic over the null space N of J
F3R_LA_FN void sqp_step(const double Om[9][9], const double r[9], double d[9]) {
  double h[6], J[6][9];
  constraints(r, h, J);
  // x = J^T (J J^T)^-1 (-h)
  double G[6][6], mh[6], z[6], x[9];
  for (int i = 0; i < 6; ++i) {
    mh[i] = -h[i];
    for (int j = 0; j < 6; ++j) {
      double a = 0;
      for (int k = 0; k < 9; ++k) a += J[i][k] * J[j][k];
      G[i][j] = a;
    }
  }
  if (!f3r_la::chol_solve<6>(G, mh, z)) {
    for (int i = 0; i < 6; ++i) G[i][i] += 1e-12;
    if (!f3r_la::chol_solve<6>(G, mh, z))
      for (int i = 0; i < 6; ++i) z[i] = 0.0;
  }
  for (int k = 0; k < 9; ++k) {
    double a = 0;
    for (int i = 0; i < 6; ++i) a += J[i][k] * z[i];
    x[k] = a;
  }
  // orthonormal basis H of the row space (modified Gram-Schmidt), projector onto its complement, three deflation steps -> N (9 x 3)
  double Pn[9][9];
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j < 9; ++j) Pn[i][j] = (i == j) ? 1.0 : 0.0;
  double Hq[6][9];
  for (int i = 0; i < 6; ++i) {
    double v[9];
    for (int k = 0; k < 9; ++k) v[k] = J[i][k];
    for (int pass = 0; pass < 2; ++pass)
      for (int j = 0; j < i; ++j) {
        double a = 0;
        for (int k = 0; k < 9; ++k) a += Hq[j][k] * v[k];
        for (int k = 0; k < 9; ++k) v[k] -= a * Hq[j][k];
      }
    double n = 0;
    for (int k = 0; k < 9; ++k) n += v[k] * v[k];
    n = sqrt(n);
    for (int k = 0; k < 9; ++k) Hq[i][k] = n > 1e-300 ? v[k] / n : 0.0;
    for (int a = 0; a < 9; ++a)
      for (int b = 0; b < 9; ++b) Pn[a][b] -= Hq[i][a] * Hq[i][b];
  }
  double N[3][9];
  for (int j = 0; j < 3; ++j) {
    int best = 0;
    double bn = -1;
    for (int c = 0; c < 9; ++c) {
      double n = 0;
      for (int k = 0; k < 9; ++k) n += Pn[k][c] * Pn[k][c];
      if (n > bn) { bn = n; best = c; }
    }
    const double inv = bn > 1e-300 ? 1.0 / sqrt(bn) : 0.0;
    for (int k = 0; k < 9; ++k) N[j][k] = Pn[k][best] * inv;
    // re-orthogonalise against the row space and the vectors already taken (round-off), then deflate
    for (int i = 0; i < 6; ++i) {
      double a = 0;
      for (int k = 0; k < 9; ++k) a += Hq[i][k] * N[j][k];
      for (int k = 0; k < 9; ++k) N[j][k] -= a * Hq[i][k];
    }
    for (int i = 0; i < j; ++i) {
      double a = 0;
      for (int k = 0; k < 9; ++k) a += N[i][k] * N[j][k];
      for (int k = 0; k < 9; ++k) N[j][k] -= a * N[i][k];
    }
    double n = 0;
    for (int k = 0; k < 9; ++k) n += N[j][k] * N[j][k];
    n = sqrt(n);
    for (int k = 0; k < 9; ++k) N[j][k] = n > 1e-300 ? N[j][k] / n : 0.0;
    for (int a = 0; a < 9; ++a)
      for (int b = 0; b < 9; ++b) Pn[a][b] -= N[j][a] * N[j][b];
  }
  // reduced problem: (N^T Omega N) y = -N^T Omega (r + x)
  double ON[3][9], rx[9], A3[3][3], b3[3], y[3];
  for (int k = 0; k < 9; ++k) rx[k] = r[k] + x[k];
  for (int j = 0; j < 3; ++j)
    for (int a = 0; a < 9; ++a) {
      double s = 0;
      for (int b = 0; b < 9; ++b) s += Om[a][b] * N[j][b];
      ON[j][a] = s;
    }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int a = 0; a < 9; ++a) s += N[i][a] * ON[j][a];
      A3[i][j] = s;
    }
    double s = 0;
    for (int a = 0; a < 9; ++a) s += ON[i][a] * rx[a];
    b3[i] = -s;
  }
  solve3(A3, b3, y);
  for (int k = 0; k < 9; ++k) d[k] = x[k] + N[0][k] * y[0] + N[1][k] * y[1] + N[2][k] * y[2];
}

F3R_LA_FN void run_sqp(const double Om[9][9], const double r0[9], double out[9]) {
  double r[9], d[9];
  for (int k = 0; k < 9; ++k) r[k] = r0[k];
  for (int it = 0; it < SQP_MAX_ITER; ++it) {
    sqp_step(Om, r, d);
    double dd = 0;
    for (int k = 0; k < 9; ++k) { r[k] += d[k]; dd += d[k] * d[k]; }
    if (dd < SQP_TOL) break;
  }
  if (det9(r) < 0)
    for (int k = 0; k < 9; ++k) r[k] = -r[k];
  nearest_rotation(r, out);
}

// sums: N_SUMS accumulated values (accumulate() over the points); unit2: the square of the world unit the points were expressed in
// (f3r_pnp.hip conditions them as (M - centroid) / sigma: Omega scales with sigma^-2, the thresholds above are stated for raw units).
// The translation returned is in the units of the points handed to accumulate().
F3R_LA_FN bool solve(const double* s, double unit2, Result& out) {
  out.ok = 0;
  const double* S1 = s;
  const double* Sx = s + 10;
  const double* Sy = s + 20;
  const double* Sq = s + 30;
  const double n = S1[0];
  if (!(n >= 3.0)) return false;
  auto MM = [](const double* S, int c, int d) {  // sum w M_c M_d
    const int lo = c < d ? c : d, hi = c < d ? d : c;
    const int idx = lo == 0 ? hi : (lo == 1 ? 2 + hi : 5);
    return S[4 + idx];
  };
  // sum Q (3 x 3), sum Q A (3 x 9), sum A^T Q A (9 x 9): block (a, b) of the latter is sum Q_ab M M^T, of Q A it is sum Q_ab M^T
  const double* Qw[3][3] = {{S1, nullptr, Sx}, {nullptr, S1, Sy}, {Sx, Sy, Sq}};
  const double Qsgn[3][3] = {{1, 0, -1}, {0, 1, -1}, {-1, -1, 1}};
  double Qs[3][3], QA[3][9], Om[9][9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      const double* W = Qw[a][b];
      Qs[a][b] = W ? Qsgn[a][b] * W[0] : 0.0;
      for (int c = 0; c < 3; ++c) {
        QA[a][3 * b + c] = W ? Qsgn[a][b] * W[1 + c] : 0.0;
        for (int d = 0; d < 3; ++d) Om[3 * a + c][3 * b + d] = W ? Qsgn[a][b] * MM(W, c, d) : 0.0;
      }
    }
  // P = -(sum Q)^-1 sum Q A  (3 x 9)
  const double detQ = f3r_la::det3(Qs);
  if (!(fabs(detQ) > 1e-300)) return false;
  double Qi[3][3];
  Qi[0][0] = (Qs[1][1] * Qs[2][2] - Qs[1][2] * Qs[2][1]) / detQ;
  Qi[0][1] = (Qs[0][2] * Qs[2][1] - Qs[0][1] * Qs[2][2]) / detQ;
  Qi[0][2] = (Qs[0][1] * Qs[1][2] - Qs[0][2] * Qs[1][1]) / detQ;
  Qi[1][0] = (Qs[1][2] * Qs[2][0] - Qs[1][0] * Qs[2][2]) / detQ;
  Qi[1][1] = (Qs[0][0] * Qs[2][2] - Qs[0][2] * Qs[2][0]) / detQ;
  Qi[1][2] = (Qs[0][2] * Qs[1][0] - Qs[0][0] * Qs[1][2]) / detQ;
  Qi[2][0] = (Qs[1][0] * Qs[2][1] - Qs[1][1] * Qs[2][0]) / detQ;
  Qi[2][1] = (Qs[0][1] * Qs[2][0] - Qs[0][0] * Qs[2][1]) / detQ;
  Qi[2][2] = (Qs[0][0] * Qs[1][1] - Qs[0][1] * Qs[1][0]) / detQ;
  double P[3][9];
  for (int a = 0; a < 3; ++a)
    for (int k = 0; k < 9; ++k) P[a][k] = -(Qi[a][0] * QA[0][k] + Qi[a][1] * QA[1][k] + Qi[a][2] * QA[2][k]);
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j < 9; ++j) Om[i][j] += QA[0][i] * P[0][j] + QA[1][i] * P[1][j] + QA[2][i] * P[2][j];
  for (int i = 0; i < 9; ++i)
    for (int j = i + 1; j < 9; ++j) Om[i][j] = Om[j][i] = 0.5 * (Om[i][j] + Om[j][i]);
  // eigen-decomposition, ascending
  double E[9][9], V[9][9];
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j < 9; ++j) E[i][j] = Om[i][j];
  f3r_la::jacobi_sym<9>(E, V);
  int order[9];
  for (int i = 0; i < 9; ++i) order[i] = i;
  for (int i = 0; i < 8; ++i)
    for (int j = i + 1; j < 9; ++j)
      if (E[order[j]][order[j]] < E[order[i]][order[i]]) { const int tmp = order[i]; order[i] = order[j]; order[j] = tmp; }
  double ev[9];
  int num_null = 0;
  for (int i = 0; i < 9; ++i) {
    ev[i] = E[order[i]][order[i]] * unit2;  // in raw world units, where the tolerances are stated
    if (ev[i] < RANK_TOL) ++num_null;
  }
  if (num_null > 6) return false;
  const double mean[3] = {S1[1] / n, S1[2] / n, S1[3] / n};
  double best_err = 1e300;
  bool have = false;
  auto handle = [&](const double* rh) {
    double t[3];
    for (int a = 0; a < 3; ++a) {
      double v = 0;
      for (int k = 0; k < 9; ++k) v += P[a][k] * rh[k];
      t[a] = v;
    }
    if (rh[6] * mean[0] + rh[7] * mean[1] + rh[8] * mean[2] + t[2] <= 0) return;  // cheirality on the centroid (authors' test)
    double err = 0;
    for (int i = 0; i < 9; ++i) {
      double v = 0;
      for (int j = 0; j < 9; ++j) v += Om[i][j] * rh[j];
      err += rh[i] * v;
    }
    err *= unit2;
    if (!have || err < best_err - EQUAL_SQ_ERR) {
      have = true;
      best_err = err;
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) out.R[i][j] = rh[i * 3 + j];
        out.t[i] = t[i];
      }
    }
  };
  auto try_vector = [&](int col) {
    double e[9], h[6], J[6][9];
    const double s3 = sqrt(3.0);
    for (int k = 0; k < 9; ++k) e[k] = s3 * V[k][order[col]];
    constraints(e, h, J);
    double hh = 0;
    for (int i = 0; i < 6; ++i) hh += h[i] * h[i];
    if (hh < ORTHO_SQ_TOL) {
      const double sg = det9(e) < 0 ? -1.0 : 1.0;
      for (int k = 0; k < 9; ++k) e[k] *= sg;
      handle(e);
      return;
    }
    for (int sgn = 0; sgn < 2; ++sgn) {
      double e2[9], r0[9], rr[9];
      for (int k = 0; k < 9; ++k) e2[k] = sgn ? -e[k] : e[k];
      nearest_rotation(e2, r0);
      run_sqp(Om, r0, rr);
      handle(rr);
    }
  };
  const int num_eigen = num_null > 0 ? num_null : 1;
  for (int i = 0; i < num_eigen; ++i) try_vector(i);
  int idx = num_eigen;
  while (idx < 9 && (!have || best_err > 3.0 * ev[idx])) {
    try_vector(idx);
    ++idx;
  }
  if (!have) return false;
  out.err = best_err;
  out.ok = 1;
  return true;
}

}  // namespace f3r_sqpnp
