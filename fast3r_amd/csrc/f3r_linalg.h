// Small dense linear algebra in fp64 for the post-processing kernels (one thread per problem): 3x3 SVD, symmetric Jacobi eigen-solver,
// Cholesky solve.  Header-only; device functions in the library -- with F3R_HOST_BUILD defined the same source compiles as plain C++
// (tests build f3r_sqpnp.h that way to check the arithmetic on the CPU; the product never does).
#pragma once
#ifdef F3R_HOST_BUILD
#include <cmath>
#define F3R_LA_FN inline
#define F3R_LA_FORCE inline
#else
#include <hip/hip_runtime.h>
#define F3R_LA_FN __device__ inline
#define F3R_LA_FORCE __device__ __forceinline__
#endif

namespace f3r_la {

// 3x3 SVD of M by two-sided use of the Jacobi eigen-decomposition of M^T M:  M = U diag(s) V^T, s sorted descending.
F3R_LA_FN void svd3(const double M[3][3], double U[3][3], double S[3], double V[3][3]) {
  double A[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += M[k][i] * M[k][j];
      A[i][j] = a;
      V[i][j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 30; ++sweep) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (fabs(A[p][q]) < 1e-300) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {  // A <- A J
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {  // A <- J^T A
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {  // V <- V J
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  double ev[3] = {A[0][0], A[1][1], A[2][2]};
  int idx[3] = {0, 1, 2};
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (ev[idx[j]] > ev[idx[i]]) { const int tmp = idx[i]; idx[i] = idx[j]; idx[j] = tmp; }
  double Vs[3][3];
  for (int c = 0; c < 3; ++c) {
    S[c] = sqrt(fmax(ev[idx[c]], 0.0));
    for (int r = 0; r < 3; ++r) Vs[r][c] = V[r][idx[c]];
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) V[r][c] = Vs[r][c];
  // U columns: M v_c / s_c; complete a deficient basis by Gram-Schmidt / cross products
  const double tol = 1e-12 * fmax(S[0], 1e-300);
  for (int c = 0; c < 3; ++c) {
    for (int r = 0; r < 3; ++r) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += M[r][k] * V[k][c];
      U[r][c] = (S[c] > tol) ? a / S[c] : 0.0;
    }
  }
  auto norm3 = [](double* v0, double* v1, double* v2) {
    const double n = sqrt(*v0 * *v0 + *v1 * *v1 + *v2 * *v2);
    if (n > 0) { *v0 /= n; *v1 /= n; *v2 /= n; }
    return n;
  };
  if (S[0] <= tol) { U[0][0] = 1; U[1][0] = 0; U[2][0] = 0; }
  if (S[1] <= tol) {  // any unit vector orthogonal to u0
    const double a0 = fabs(U[0][0]), a1 = fabs(U[1][0]), a2 = fabs(U[2][0]);
    double e[3] = {0, 0, 0};
    e[(a0 <= a1 && a0 <= a2) ? 0 : (a1 <= a2 ? 1 : 2)] = 1.0;
    const double d = e[0] * U[0][0] + e[1] * U[1][0] + e[2] * U[2][0];
    U[0][1] = e[0] - d * U[0][0]; U[1][1] = e[1] - d * U[1][0]; U[2][1] = e[2] - d * U[2][0];
    norm3(&U[0][1], &U[1][1], &U[2][1]);
  }
  if (S[2] <= tol) {  // u2 = u0 x u1 (the sign is fixed by the determinant correction of the caller)
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
  }
}

F3R_LA_FORCE double det3(const double A[3][3]) {
  return A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
         A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
}


// Eigen-decomposition of a symmetric N x N matrix by cyclic Jacobi rotations: A <- diag(eigenvalues), V columns = eigenvectors.
template <int N>
F3R_LA_FN void jacobi_sym(double A[N][N], double V[N][N]) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0, dia = 0.0;
    for (int i = 0; i < N; ++i) {
      dia += fabs(A[i][i]);
      for (int j = i + 1; j < N; ++j) off += fabs(A[i][j]);
    }
    if (off <= 1e-18 * dia || off < 1e-300) break;
    for (int p = 0; p < N - 1; ++p)
      for (int q = p + 1; q < N; ++q) {
        if (fabs(A[p][q]) < 1e-300) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < N; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < N; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < N; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
}

// Solve A x = b for a symmetric positive definite N x N matrix (Cholesky); returns false when A is not numerically SPD.
template <int N>
F3R_LA_FN bool chol_solve(const double A[N][N], const double* b, double* x) {
  double L[N][N];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[i][j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(s > 0.0)) return false;
        L[i][i] = sqrt(s);
      } else {
        L[i][j] = s / L[j][j];
      }
    }
  double y[N];
  for (int i = 0; i < N; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
    y[i] = s / L[i][i];
  }
  for (int i = N - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < N; ++k) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
  return true;
}

}  // namespace f3r_la
