// Camera pose of every view from its global pointmap (second half): the pose step of
// MultiViewDUSt3RLitModule.estimate_camera_poses (fast3r/models/multiview_dust3r_module.py:807-869 -> :1038-1078 -> fast_pnp,
// fast3r/dust3r/cloud_opt/init_im_poses.py:300-350).  The reference hands every view (and, when the focal is unknown, every one of
// 100 tentative focals) to cv2.solvePnPRansac(SQPNP) in a CPU thread pool; here a view is one 1024-thread workgroup:
//   A. mask = conf > thr (:1045), centroid / RMS of the masked world points (conditioning)
//   B. 32 deterministic 6-point samples -> closed-form calibrated DLT each (one thread per sample) -> every hypothesis scored on ALL
//      points in one pass (inliers at 5 px, truncated cost)              [the role RANSAC plays in the reference]
//   C. moments of the inliers of the best hypothesis (one pass, 40 fp64 sums) -> closed-form DLT.  The DLT is uncalibrated in
//      disguise: with P = [X 1] and centred pixels (px, py) the normal matrix of {r1.P - (px/f) r3.P = 0, r2.P - (py/f) r3.P = 0} is
//      [[S0,0,-S1x/f],[0,S0,-S1y/f],[.,.,S2/f^2]]; eliminating the first two blocks leaves (S2 - S1x S0^-1 S1x - S1y S0^-1 S1y) c =
//      lambda c for c = [r3 t3] independently of f, and [r1 t1] = S0^-1 S1x c / f, [r2 t2] = S0^-1 S1y c / f: one 4x4 eigenproblem
//      serves every focal, and |r1| / |r3| estimates the focal itself
//   D. focal: the given one, or the candidate of np.geomspace(S/2, 3S, 100) (:312-316) nearest to the DLT estimate and its two
//      neighbours, each refined and scored (inlier count :342, ties by truncated cost)
//   E. nearest rotation (3x3 SVD), 6 gated Gauss-Newton steps on the reprojection error (28 fp64 sums per pass)
//   F. SQPnP (f3r_sqpnp.h) on the points within 5 px of the selected pose: the solver the reference names, on the consensus set, as
//      OpenCV's solvePnPRansac ends; cam-to-world (:349-350)
// Same algorithm, independently written on torch.linalg: oracle/pnp_oracle.py.  All passes re-read the view's 4 MB from L2.
#include "f3r_common.h"
#include "f3r_linalg.h"
#include "f3r_sqpnp.h"

namespace {

constexpr int PT = 1024;
constexpr int NW_ = PT / 64;
constexpr int N_HYP = 32, SAMPLE = 6, N_GN = 6;
constexpr float THR2 = 25.0f;  // reprojectionError = 5 px (init_im_poses.py:335)

// out of line: the solver's 9 x 9 temporaries live in its own frame instead of lengthening the live ranges of the kernel's hot loops
__device__ __attribute__((noinline)) bool sqpnp_solve(const double* sums, double unit2, f3r_sqpnp::Result* out) {
  return f3r_sqpnp::solve(sums, unit2, *out);
}

struct Pose {  // world -> camera
  double R[3][3];
  double t[3];
  double f;
  int ok;
};

// block sum of K doubles held per thread in v[]; result broadcast to every thread through out[]
template <int K>
__device__ void block_sum(double* v, double (*red)[48], double* out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double a = v[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) red[wv][k] = a;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double s = 0.0;
    for (int w = 0; w < NW_; ++w) s += red[w][threadIdx.x];
    out[threadIdx.x] = s;
  }
  __syncthreads();
}

// moments layout (40): S0 (10 upper-triangular entries of sum P P^T), S1x (10), S1y (10), S2 (10); P = [xn yn zn 1]
__device__ __forceinline__ void add_moments(double* m, const double P[4], double px, double py) {
  const double w2 = px * px + py * py;
  int k = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = i; j < 4; ++j) {
      const double pp = P[i] * P[j];
      m[k] += pp;
      m[10 + k] += px * pp;
      m[20 + k] += py * pp;
      m[30 + k] += w2 * pp;
      ++k;
    }
}

__device__ void unpack_sym4(const double* m, double S[4][4]) {
  int k = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = i; j < 4; ++j) {
      S[i][j] = m[k];
      S[j][i] = m[k];
      ++k;
    }
}

__device__ double det_sym4(const double S[4][4]) {  // determinant of an SPD candidate through its Cholesky factor (0 when not SPD)
  double L[4][4];
  double det = 1.0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = S[i][j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(s > 0.0)) return 0.0;
        L[i][i] = sqrt(s);
        det *= s;
      } else {
        L[i][j] = s / L[j][j];
      }
    }
  return det;
}

// closed-form DLT rows from the 40 moments: a, b (still to be divided by the focal) and c; false when S0 is singular
__device__ bool dlt_rows(const double* m, double a[4], double b[4], double c[4], double min_det = 0.0) {
  double S0[4][4], S1x[4][4], S1y[4][4], S2[4][4];
  unpack_sym4(m, S0);
  if (min_det > 0.0 && det_sym4(S0) < min_det) return false;
  unpack_sym4(m + 10, S1x);
  unpack_sym4(m + 20, S1y);
  unpack_sym4(m + 30, S2);
  // X = S0^-1 S1x, Y = S0^-1 S1y column by column
  double X[4][4], Y[4][4];
  for (int col = 0; col < 4; ++col) {
    double bx[4], by[4], sx[4], sy[4];
    for (int i = 0; i < 4; ++i) { bx[i] = S1x[i][col]; by[i] = S1y[i][col]; }
    if (!f3r_la::chol_solve<4>(S0, bx, sx) || !f3r_la::chol_solve<4>(S0, by, sy)) return false;
    for (int i = 0; i < 4; ++i) { X[i][col] = sx[i]; Y[i][col] = sy[i]; }
  }
  double G[4][4], V[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double g = S2[i][j];
      for (int k = 0; k < 4; ++k) g -= S1x[i][k] * X[k][j] + S1y[i][k] * Y[k][j];
      G[i][j] = g;
    }
  for (int i = 0; i < 4; ++i)
    for (int j = i + 1; j < 4; ++j) { const double s = 0.5 * (G[i][j] + G[j][i]); G[i][j] = s; G[j][i] = s; }
  f3r_la::jacobi_sym<4>(G, V);
  int kmin = 0;
  for (int k = 1; k < 4; ++k)
    if (G[k][k] < G[kmin][kmin]) kmin = k;
  for (int i = 0; i < 4; ++i) c[i] = V[i][kmin];
  for (int i = 0; i < 4; ++i) {
    double sa = 0, sb = 0;
    for (int k = 0; k < 4; ++k) { sa += X[i][k] * c[k]; sb += Y[i][k] * c[k]; }
    a[i] = sa;
    b[i] = sb;
  }
  double depth = 0;  // sum of the depths of the points that built the moments = c . S0[:, 3]
  for (int i = 0; i < 4; ++i) depth += c[i] * S0[i][3];
  if (depth < 0)
    for (int i = 0; i < 4; ++i) { a[i] = -a[i]; b[i] = -b[i]; c[i] = -c[i]; }
  return true;
}

// rows -> proper rotation + translation (normalised world frame), then the normalisation (centroid cen, scale sig) undone
__device__ void pose_from_rows(const double a[4], const double b[4], const double c[4], double f, const double cen[3], double sig, Pose& P) {
  double A[3][3], U[3][3], S[3], V[3][3];
  for (int j = 0; j < 3; ++j) { A[0][j] = a[j] / f; A[1][j] = b[j] / f; A[2][j] = c[j]; }
  f3r_la::svd3(A, U, S, V);
  const double d = (f3r_la::det3(U) * f3r_la::det3(V) < 0) ? -1.0 : 1.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) P.R[i][j] = U[i][0] * V[j][0] + U[i][1] * V[j][1] + d * U[i][2] * V[j][2];
  const double s = (S[0] + S[1] + d * S[2]) / 3.0;
  const double tn[3] = {a[3] / f / s, b[3] / f / s, c[3] / s};
  for (int i = 0; i < 3; ++i) P.t[i] = sig * tn[i] - (P.R[i][0] * cen[0] + P.R[i][1] * cen[1] + P.R[i][2] * cen[2]);
  P.f = f;
  P.ok = (s > 0.0 && s == s) ? 1 : 0;
}

__device__ __forceinline__ uint32_t sample_index(int h, int j, uint32_t npix) {
  uint32_t x = (1103515245u * (uint32_t)(h * SAMPLE + j + 1) + 12345u) & 0x7FFFFFFFu;
  x = x * 2654435761u;
  return x % npix;
}

__device__ __forceinline__ double grid_focal(int k, int S, int n) {  // np.geomspace(S / 2, 3 S, n)[k]
  return n > 1 ? 0.5 * S * exp(log(6.0) * (double)k / (double)(n - 1)) : 0.5 * S;
}

__global__ __launch_bounds__(PT) void pnp_kernel(const float* __restrict__ pts, const float* __restrict__ conf, const float* __restrict__ focal_in,
                                                 float* __restrict__ focal_out, float* __restrict__ pose_out, int* __restrict__ inliers_out,
                                                 int H, int W, float conf_thr, float ppx, float ppy, int n_focals, int n_hyp) {
  __shared__ double red[NW_][48];
  __shared__ double sums[48];
  __shared__ Pose hyp[N_HYP];
  __shared__ Pose cur;
  __shared__ double sh_rows[12];
  __shared__ int sh_best;
  const int64_t npix = (int64_t)H * W;
  const int64_t prob = blockIdx.x;
  const float* cf = conf + prob * npix;
  const float* pt = pts + prob * npix * 3;
  const int tid = threadIdx.x;
  float* pose_o = pose_out + prob * 16;

  auto fail = [&]() {
    if (tid == 0) {
      for (int i = 0; i < 16; ++i) pose_o[i] = (i % 5 == 0) ? 1.f : 0.f;  // identity (multiview_dust3r_module.py:1062-1064)
      focal_out[prob] = __builtin_nanf("");
      inliers_out[prob] = 0;
    }
  };

  // ---- A. masked count, centroid, RMS
  double acc[48];
  for (int k = 0; k < 5; ++k) acc[k] = 0.0;
  for (int64_t i = tid; i < npix; i += PT)
    if (cf[i] > conf_thr) {
      const double x = pt[i * 3], y = pt[i * 3 + 1], z = pt[i * 3 + 2];
      acc[0] += 1.0; acc[1] += x; acc[2] += y; acc[3] += z; acc[4] += x * x + y * y + z * z;
    }
  block_sum<5>(acc, red, sums);
  const double n_mask = sums[0];
  if (n_mask < 4.0) {  // init_im_poses.py:302-303
    fail();
    return;
  }
  const double cen[3] = {sums[1] / n_mask, sums[2] / n_mask, sums[3] / n_mask};
  const double var = sums[4] / n_mask - (cen[0] * cen[0] + cen[1] * cen[1] + cen[2] * cen[2]);
  const double sig = fmax(sqrt(fmax(var, 0.0)), 1e-12);
  const double isig = 1.0 / sig;
  const int Smax = H > W ? H : W;
  const bool focal_known = focal_in != nullptr && focal_in[prob] > 0.f;
  __syncthreads();

  // ---- B. sampled hypotheses (one thread each), then scored on all points
  if (tid < N_HYP) {
    Pose P;
    P.ok = 0;
    P.f = 0;
    if (n_mask >= SAMPLE && tid < n_hyp) {  // n_hyp = the caller's RANSAC iteration count (niter_PnP), at most N_HYP
      double m[40];
      for (int k = 0; k < 40; ++k) m[k] = 0.0;
      for (int j = 0; j < SAMPLE; ++j) {
        uint32_t i = sample_index(tid, j, (uint32_t)npix);
        while (!(cf[i] > conf_thr)) i = (i + 1 == (uint32_t)npix) ? 0u : i + 1;
        const double Pn[4] = {(pt[(int64_t)i * 3] - cen[0]) * isig, (pt[(int64_t)i * 3 + 1] - cen[1]) * isig, (pt[(int64_t)i * 3 + 2] - cen[2]) * isig, 1.0};
        add_moments(m, Pn, (double)(int)(i % (uint32_t)W) - ppx, (double)(int)(i / (uint32_t)W) - ppy);
      }
      double a[4], b[4], c[4];
      if (dlt_rows(m, a, b, c, 1e-12)) {  // degenerate (coplanar / repeated) samples are skipped
        const double nc = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        if (nc > 1e-12) {
          double f;
          if (focal_known) {
            f = focal_in[prob];
          } else {
            const double fd = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) / nc;
            int k = (int)llrint(log(fmax(fd, 1e-9) / (0.5 * Smax)) / log(6.0) * (n_focals - 1));
            k = k < 0 ? 0 : (k > n_focals - 1 ? n_focals - 1 : k);
            f = grid_focal(k, Smax, n_focals);
          }
          pose_from_rows(a, b, c, f, cen, sig, P);
        }
      }
    }
    hyp[tid] = P;
  }
  __syncthreads();
  // score: per hypothesis inlier count and truncated cost, 8 hypotheses per pass (16 accumulators)
  int best_h = -1;
  {
    double best_cnt = 0.0, best_cost = 1e300;
    for (int h0 = 0; h0 < N_HYP; h0 += 8) {
      for (int k = 0; k < 16; ++k) acc[k] = 0.0;
      for (int64_t i = tid; i < npix; i += PT) {
        if (!(cf[i] > conf_thr)) continue;
        const double x = pt[i * 3], y = pt[i * 3 + 1], z = pt[i * 3 + 2];
        const double px = (double)(int)(i % W) - ppx, py = (double)(int)(i / W) - ppy;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const Pose& P = hyp[h0 + q];
          const double zc = P.R[2][0] * x + P.R[2][1] * y + P.R[2][2] * z + P.t[2];
          const double xc = P.R[0][0] * x + P.R[0][1] * y + P.R[0][2] * z + P.t[0];
          const double yc = P.R[1][0] * x + P.R[1][1] * y + P.R[1][2] * z + P.t[1];
          const double ex = P.f * xc / zc - px, ey = P.f * yc / zc - py;
          const double e2 = ex * ex + ey * ey;
          const bool inl = P.ok && zc > 0.0 && e2 <= (double)THR2;
          acc[2 * q] += inl ? 1.0 : 0.0;
          acc[2 * q + 1] += inl ? e2 : (double)THR2;
        }
      }
      block_sum<16>(acc, red, sums);
      for (int q = 0; q < 8; ++q)
        if (hyp[h0 + q].ok && (sums[2 * q] > best_cnt || (sums[2 * q] == best_cnt && sums[2 * q] > 0 && sums[2 * q + 1] < best_cost))) {
          best_cnt = sums[2 * q];
          best_cost = sums[2 * q + 1];
          best_h = h0 + q;
        }
      __syncthreads();
    }
    if (best_cnt < SAMPLE) best_h = -1;
  }

  // ---- C. moments over the inliers of the best hypothesis (all masked points when there is none) -> DLT rows
  {
    for (int k = 0; k < 40; ++k) acc[k] = 0.0;
    Pose P;
    if (best_h >= 0) P = hyp[best_h];
    for (int64_t i = tid; i < npix; i += PT) {
      if (!(cf[i] > conf_thr)) continue;
      const double x = pt[i * 3], y = pt[i * 3 + 1], z = pt[i * 3 + 2];
      const double px = (double)(int)(i % W) - ppx, py = (double)(int)(i / W) - ppy;
      if (best_h >= 0) {
        const double zc = P.R[2][0] * x + P.R[2][1] * y + P.R[2][2] * z + P.t[2];
        const double xc = P.R[0][0] * x + P.R[0][1] * y + P.R[0][2] * z + P.t[0];
        const double yc = P.R[1][0] * x + P.R[1][1] * y + P.R[1][2] * z + P.t[1];
        const double ex = P.f * xc / zc - px, ey = P.f * yc / zc - py;
        if (!(zc > 0.0 && ex * ex + ey * ey <= (double)THR2)) continue;
      }
      const double Pn[4] = {(x - cen[0]) * isig, (y - cen[1]) * isig, (z - cen[2]) * isig, 1.0};
      add_moments(acc, Pn, px, py);
    }
    block_sum<40>(acc, red, sums);
  }
  if (tid == 0) {
    double m[40], a[4], b[4], c[4];
    for (int k = 0; k < 40; ++k) m[k] = sums[k];
    sh_best = dlt_rows(m, a, b, c) ? 1 : 0;
    for (int k = 0; k < 4; ++k) { sh_rows[k] = a[k]; sh_rows[4 + k] = b[k]; sh_rows[8 + k] = c[k]; }
  }
  __syncthreads();
  if (!sh_best) {
    fail();
    return;
  }

  // ---- D / E. candidate focals, each: pose from rows -> gated Gauss-Newton -> score
  int k_lo = 0, k_hi = 0;
  if (!focal_known) {
    const double* a = sh_rows;
    const double* b = sh_rows + 4;
    const double* c = sh_rows + 8;
    const double nc = fmax(sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]), 1e-30);
    const double fd = sqrt(fmax(sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]), 1e-30)) / nc;
    int k0 = (int)llrint(log(fmax(fd, 1e-9) / (0.5 * Smax)) / log(6.0) * (n_focals - 1));
    k0 = k0 < 0 ? 0 : (k0 > n_focals - 1 ? n_focals - 1 : k0);
    k_lo = k0 - 1 < 0 ? 0 : k0 - 1;
    k_hi = k0 + 1 > n_focals - 1 ? n_focals - 1 : k0 + 1;
  }
  double best_cnt = 0.0, best_cost = 1e300;
  Pose best_pose;
  best_pose.ok = 0;
  for (int kc = k_lo; kc <= k_hi; ++kc) {
    const double f = focal_known ? (double)focal_in[prob] : grid_focal(kc, Smax, n_focals);
    if (tid == 0) pose_from_rows(sh_rows, sh_rows + 4, sh_rows + 8, f, cen, sig, cur);
    __syncthreads();
    for (int it = 0; it <= N_GN; ++it) {  // N_GN refinement passes, then one scoring pass
      const Pose P = cur;
      for (int k = 0; k < 30; ++k) acc[k] = 0.0;
      for (int64_t i = tid; i < npix; i += PT) {
        if (!(cf[i] > conf_thr)) continue;
        const double X = pt[i * 3], Y = pt[i * 3 + 1], Z = pt[i * 3 + 2];
        const double px = (double)(int)(i % W) - ppx, py = (double)(int)(i / W) - ppy;
        const double z = P.R[2][0] * X + P.R[2][1] * Y + P.R[2][2] * Z + P.t[2];
        const double x = P.R[0][0] * X + P.R[0][1] * Y + P.R[0][2] * Z + P.t[0];
        const double y = P.R[1][0] * X + P.R[1][1] * Y + P.R[1][2] * Z + P.t[1];
        const double iz = 1.0 / z;
        const double ex = f * x * iz - px, ey = f * y * iz - py;
        const double e2 = ex * ex + ey * ey;
        const bool inl = z > 0.0 && e2 <= (double)THR2;
        acc[28] += inl ? 1.0 : 0.0;
        acc[29] += inl ? e2 : (double)THR2;
        if (!inl || it == N_GN) continue;
        const double Jx[6] = {-f * x * y * iz * iz, f * (1.0 + x * x * iz * iz), -f * y * iz, f * iz, 0.0, -f * x * iz * iz};
        const double Jy[6] = {-f * (1.0 + y * y * iz * iz), f * x * y * iz * iz, f * x * iz, 0.0, f * iz, -f * y * iz * iz};
        int k = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
          for (int c = r; c < 6; ++c) acc[k++] += Jx[r] * Jx[c] + Jy[r] * Jy[c];
          acc[21 + r] += Jx[r] * ex + Jy[r] * ey;
        }
      }
      block_sum<30>(acc, red, sums);
      if (it == N_GN) break;
      if (tid == 0 && sums[28] >= 4.0) {
        double Hm[6][6], g[6], d[6];
        int k = 0;
        for (int r = 0; r < 6; ++r)
          for (int c = r; c < 6; ++c) { Hm[r][c] = sums[k]; Hm[c][r] = sums[k]; ++k; }
        for (int r = 0; r < 6; ++r) { Hm[r][r] += 1e-9 * Hm[r][r] + 1e-12; g[r] = -sums[21 + r]; }
        if (f3r_la::chol_solve<6>(Hm, g, d)) {
          // dR = exp([w]_x) (Rodrigues), R <- dR R, t <- dR t + dt
          const double th = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
          const double K[3][3] = {{0, -d[2], d[1]}, {d[2], 0, -d[0]}, {-d[1], d[0], 0}};
          const double c1 = th < 1e-12 ? 1.0 : sin(th) / th, c2 = th < 1e-12 ? 0.0 : (1.0 - cos(th)) / (th * th);
          double dR[3][3], Rn[3][3], tn[3];
          for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
              double kk = 0;
              for (int q = 0; q < 3; ++q) kk += K[r][q] * K[q][c];
              dR[r][c] = (r == c ? 1.0 : 0.0) + c1 * K[r][c] + c2 * kk;
            }
          for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) Rn[r][c] = dR[r][0] * cur.R[0][c] + dR[r][1] * cur.R[1][c] + dR[r][2] * cur.R[2][c];
            tn[r] = dR[r][0] * cur.t[0] + dR[r][1] * cur.t[1] + dR[r][2] * cur.t[2] + d[3 + r];
          }
          for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) cur.R[r][c] = Rn[r][c];
            cur.t[r] = tn[r];
          }
        }
      }
      __syncthreads();
    }
    const double cnt = sums[28], cost = sums[29];
    if (cnt > best_cnt || (cnt == best_cnt && cnt > 0 && cost < best_cost)) {
      best_cnt = cnt;
      best_cost = cost;
      best_pose = cur;
      best_pose.f = f;
      best_pose.ok = 1;
    }
    __syncthreads();
    if (focal_known) break;
  }
  if (!best_pose.ok || best_cnt < 1.0) {
    fail();
    return;
  }
  // ---- F. SQPnP on the consensus set: cv2.solvePnPRansac(..., flags=SOLVEPNP_SQPNP) (init_im_poses.py:335) ends with the named solver on
  // its inliers, so that -- not the Gauss-Newton iterate that selected them -- is the pose the reference returns.  One pass of 40 fp64
  // sums over the points within 5 px of the selected pose (conditioned world points, pixels divided by the focal), then f3r_sqpnp::solve
  // on one thread; the selected pose stays if the solver finds no admissible candidate.
  {
    for (int k = 0; k < f3r_sqpnp::N_SUMS; ++k) acc[k] = 0.0;
    const Pose P = best_pose;
    const double f = P.f, inv_f = 1.0 / P.f;
    for (int64_t i = tid; i < npix; i += PT) {
      if (!(cf[i] > conf_thr)) continue;
      const double X = pt[i * 3], Y = pt[i * 3 + 1], Z = pt[i * 3 + 2];
      const double px = (double)(int)(i % W) - ppx, py = (double)(int)(i / W) - ppy;
      const double z = P.R[2][0] * X + P.R[2][1] * Y + P.R[2][2] * Z + P.t[2];
      const double x = P.R[0][0] * X + P.R[0][1] * Y + P.R[0][2] * Z + P.t[0];
      const double y = P.R[1][0] * X + P.R[1][1] * Y + P.R[1][2] * Z + P.t[1];
      const double ex = f * x / z - px, ey = f * y / z - py;
      if (!(z > 0.0 && ex * ex + ey * ey <= (double)THR2)) continue;
      const double Mn[3] = {(X - cen[0]) * isig, (Y - cen[1]) * isig, (Z - cen[2]) * isig};
      f3r_sqpnp::accumulate(acc, Mn, px * inv_f, py * inv_f);
    }
    block_sum<f3r_sqpnp::N_SUMS>(acc, red, sums);
    if (tid == 0) {
      f3r_sqpnp::Result r;
      cur = P;
      if (sqpnp_solve(sums, sig * sig, &r)) {
        for (int a = 0; a < 3; ++a) {
          for (int b = 0; b < 3; ++b) cur.R[a][b] = r.R[a][b];
          cur.t[a] = sig * r.t[a] - (r.R[a][0] * cen[0] + r.R[a][1] * cen[1] + r.R[a][2] * cen[2]);
        }
      }
    }
    __syncthreads();
    const double keep_f = best_pose.f;
    best_pose = cur;
    best_pose.f = keep_f;
  }
  if (tid == 0) {  // cam-to-world = inverse of [R | t] (init_im_poses.py:349-350)
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) pose_o[r * 4 + c] = (float)best_pose.R[c][r];
      pose_o[r * 4 + 3] = (float)(-(best_pose.R[0][r] * best_pose.t[0] + best_pose.R[1][r] * best_pose.t[1] + best_pose.R[2][r] * best_pose.t[2]));
    }
    pose_o[12] = 0.f; pose_o[13] = 0.f; pose_o[14] = 0.f; pose_o[15] = 1.f;
    focal_out[prob] = (float)best_pose.f;
    inliers_out[prob] = (int)best_cnt;
  }
}

}  // namespace

extern "C" int f3r_estimate_poses(const float* pts3d, const float* conf, const float* focal_in, float* focal_out, float* cam_to_world,
                                  int* inliers, int n_views, int H, int W, float conf_thr, float ppx, float ppy, int n_focals, int n_iter,
                                  f3r_stream_t stream) {
  F3R_REQUIRE(pts3d && conf && focal_out && cam_to_world && inliers, "f3r_estimate_poses: null pointer");
  F3R_REQUIRE(n_views >= 0 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31), "f3r_estimate_poses: bad sizes");
  F3R_REQUIRE(n_focals >= 1 && n_focals <= 4096, "f3r_estimate_poses: n_focals %d", n_focals);
  F3R_REQUIRE(n_iter >= 1, "f3r_estimate_poses: n_iter %d (the RANSAC iteration count: at least 1)", n_iter);
  const int n_hyp = n_iter < N_HYP ? n_iter : N_HYP;
  if (n_views == 0) return F3R_OK;
  hipLaunchKernelGGL(pnp_kernel, dim3(n_views), dim3(PT), 0, (hipStream_t)stream, pts3d, conf, focal_in, focal_out, cam_to_world, inliers, H, W,
                     conf_thr, ppx, ppy, n_focals, n_hyp);
  return f3r_check_launch("f3r_estimate_poses");
}
