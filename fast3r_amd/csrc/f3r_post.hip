// Post-forward alignment of the local pointmaps to the global frame:
// MultiViewDUSt3RLitModule.align_local_pts3d_to_global, fast3r/models/multiview_dust3r_module.py:427-549.
// Per (view, sample) "problem" of npix pixels:
//   1. thr = torch.quantile(conf, q)                       -> exact order statistics by 3-pass radix select + fp32 lerp
//   2. mask = (conf >= thr) & valid; fall back to valid if fewer than 3 points survive, to identity if still fewer (:495-510)
//   3. (R, t, s) = roma.rigid_points_registration(local[mask], global[mask], compute_scaling=True): Umeyama -> masked raw
//      moments in fp64 (one pass), then a 3x3 SVD by Jacobi rotations (one thread per problem)
//   4. out = s * (local @ R^T) + t for ALL pixels (:514)
// Everything here is HBM-bound byte / index work plus a 3x3 solve: no MFMA.  One workgroup per problem for steps 1-3 (a problem is
// 1-3 MB; hundreds of problems run side by side), a flat grid for step 4.
#include "f3r_common.h"

#include "f3r_linalg.h"

namespace {

constexpr int PNT = 1024;  // threads per problem workgroup

// float -> unsigned key with the same ordering (handles negatives too; conf is >= vmin > 0 in practice)
__device__ __forceinline__ uint32_t fkey(float f) {
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __builtin_bit_cast(float, u);
}

// k-th smallest key (0-based) of conf[0..n) by radix select over 11 + 11 + 10 bits; all threads return the same value.
__device__ uint32_t select_kth(const float* __restrict__ conf, int64_t n, int64_t k, uint32_t* hist /*2048*/, int64_t* sh_i64 /*2*/) {
  uint32_t prefix = 0, prefix_mask = 0;
  const int shifts[3] = {21, 10, 0};
  const int bits[3] = {11, 11, 10};
  for (int pass = 0; pass < 3; ++pass) {
    const int nb = 1 << bits[pass];
    for (int i = threadIdx.x; i < nb; i += PNT) hist[i] = 0;
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += PNT) {
      const uint32_t key = fkey(conf[i]);
      if ((key & prefix_mask) == prefix) atomicAdd(&hist[(key >> shifts[pass]) & (nb - 1)], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int64_t acc = 0;
      int b = 0;
      for (; b < nb; ++b) {
        if (acc + hist[b] > k) break;
        acc += hist[b];
      }
      sh_i64[0] = b;
      sh_i64[1] = k - acc;
    }
    __syncthreads();
    const uint32_t b = (uint32_t)sh_i64[0];
    k = sh_i64[1];
    prefix |= b << shifts[pass];
    prefix_mask |= (uint32_t)(nb - 1) << shifts[pass];
    __syncthreads();
  }
  return prefix;
}

// workspace per problem: 40 doubles: [0..16] moments of mask A (conf & valid), [17..33] of mask B (valid), [34] thr (as double)
constexpr int WS_PER = 40;

__global__ __launch_bounds__(PNT) void align_stats_kernel(const float* __restrict__ conf, const float* __restrict__ loc,
                                                          const float* __restrict__ glob, const uint8_t* __restrict__ valid,
                                                          double* __restrict__ ws, int64_t npix, float q) {
  __shared__ uint32_t hist[2048];
  __shared__ int64_t sh_i64[2];
  __shared__ double red[16][34];
  __shared__ float sh_thr;
  const int64_t prob = blockIdx.x;
  const float* cf = conf + prob * npix;
  const float* pl = loc + prob * npix * 3;
  const float* pg = glob + prob * npix * 3;
  const uint8_t* vm = valid ? valid + prob * npix : nullptr;

  // ---- 1. quantile, exactly as torch.quantile (linear): rank = q*(n-1) in fp32, lerp between the two order statistics
  const float rank = q * (float)(npix - 1);
  const float rlo = floorf(rank);
  const int64_t klo = (int64_t)rlo;
  const int64_t khi = (int64_t)ceilf(rank);
  const float vlo = fkey_inv(select_kth(cf, npix, klo, hist, sh_i64));
  float vhi = vlo;
  if (khi != klo) vhi = fkey_inv(select_kth(cf, npix, khi, hist, sh_i64));
  if (threadIdx.x == 0) {
    const float w = rank - rlo;
    const float d = vhi - vlo;
    sh_thr = (w < 0.5f) ? vlo + w * d : vhi - d * (1.0f - w);  // at::lerp
  }
  __syncthreads();
  const float thr = sh_thr;

  // ---- 2./3. masked raw moments in fp64: n, sum x(3), sum y(3), sum y_i x_j (9), sum |x|^2   for both masks
  double mA[17], mB[17];
#pragma unroll
  for (int i = 0; i < 17; ++i) { mA[i] = 0.0; mB[i] = 0.0; }
  for (int64_t i = threadIdx.x; i < npix; i += PNT) {
    const bool v = vm ? (vm[i] != 0) : true;
    if (!v) continue;
    const bool a = cf[i] >= thr;
    const double x0 = pl[i * 3 + 0], x1 = pl[i * 3 + 1], x2 = pl[i * 3 + 2];
    const double y0 = pg[i * 3 + 0], y1 = pg[i * 3 + 1], y2 = pg[i * 3 + 2];
    const double t[17] = {1.0, x0, x1, x2, y0, y1, y2, y0 * x0, y0 * x1, y0 * x2, y1 * x0, y1 * x1, y1 * x2, y2 * x0, y2 * x1, y2 * x2,
                          x0 * x0 + x1 * x1 + x2 * x2};
#pragma unroll
    for (int j = 0; j < 17; ++j) {
      mB[j] += t[j];
      if (a) mA[j] += t[j];
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 17; ++j) {
    double a = mA[j], b = mB[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a += __shfl_xor(a, off, 64);
      b += __shfl_xor(b, off, 64);
    }
    if (lane == 0) { red[wv][j] = a; red[wv][17 + j] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 34) {
    double s = 0.0;
    for (int w = 0; w < PNT / 64; ++w) s += red[w][threadIdx.x];
    ws[prob * WS_PER + threadIdx.x] = s;
  }
  if (threadIdx.x == 0) ws[prob * WS_PER + 34] = (double)thr;
}

// one thread per problem: moments -> (R, t, s) as 13 floats [R row-major (9) | t (3) | s]
__global__ void align_solve_kernel(const double* __restrict__ ws, float* __restrict__ rts, int n_prob) {
  const int prob = blockIdx.x * blockDim.x + threadIdx.x;
  if (prob >= n_prob) return;
  const double* m = ws + (int64_t)prob * WS_PER;
  if (m[0] < 3.0) m += 17;  // fewer than 3 confident points: use the valid mask only (:495-503)
  float* o = rts + prob * 13;
  if (m[0] < 3.0) {  // identity (:506-510)
    for (int i = 0; i < 9; ++i) o[i] = (i % 4 == 0) ? 1.f : 0.f;
    o[9] = o[10] = o[11] = 0.f;
    o[12] = 1.f;
    return;
  }
  const double n = m[0];
  const double xm[3] = {m[1] / n, m[2] / n, m[3] / n}, ym[3] = {m[4] / n, m[5] / n, m[6] / n};
  double M[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[i][j] = m[7 + i * 3 + j] - n * ym[i] * xm[j];
  const double sx2 = m[16] - n * (xm[0] * xm[0] + xm[1] * xm[1] + xm[2] * xm[2]);
  double U[3][3], S[3], V[3][3];
  f3r_la::svd3(M, U, S, V);
  const double d = (f3r_la::det3(U) * f3r_la::det3(V) < 0) ? -1.0 : 1.0;
  double R[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = U[i][0] * V[j][0] + U[i][1] * V[j][1] + d * U[i][2] * V[j][2];
  const double scale = (S[0] + S[1] + d * S[2]) / sx2;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) o[i * 3 + j] = (float)R[i][j];
    o[9 + i] = (float)(ym[i] - scale * (R[i][0] * xm[0] + R[i][1] * xm[1] + R[i][2] * xm[2]));
  }
  o[12] = (float)scale;
}

// out = s * (x R^T) + t, fp32, the reference's own operation order (scale the rotated point, then translate)
__global__ void align_apply_kernel(const float* __restrict__ loc, const float* __restrict__ rts, float* __restrict__ out, int64_t npix,
                                   int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float* r = rts + (i / npix) * 13;
  const float x0 = loc[i * 3 + 0], x1 = loc[i * 3 + 1], x2 = loc[i * 3 + 2];
  const float s = r[12];
#pragma unroll
  for (int c = 0; c < 3; ++c) out[i * 3 + c] = s * (x0 * r[c * 3 + 0] + x1 * r[c * 3 + 1] + x2 * r[c * 3 + 2]) + r[9 + c];
}

__global__ void align_thr_kernel(const double* __restrict__ ws, float* __restrict__ o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = (float)ws[(int64_t)i * WS_PER + 34];
}


// ------------------------------------------------------------------------------------------------------------------------
// Focal estimation (first half): estimate_focal, fast3r/models/multiview_dust3r_module.py:1081-1109,
// = torch.quantile threshold on the confidence + the "weiszfeld" branch of estimate_focal_knowing_depth_and_confidence_mask,
// fast3r/dust3r/post_process.py:77-142 (closed-form L2 start, then n_iter re-weighted least-squares steps).  One 1024-thread
// workgroup per view: pass 0 writes (px, py, x/z, y/z) of the points above the threshold into the workspace (zeros elsewhere:
// such a point weighs 1e8 on two zero terms), every iteration is then one pass over that 16-byte-per-pixel image (it stays in L2)
// with fp32 per-point arithmetic in the reference's operation order (no fma contraction) and fp64 block sums.
__device__ __forceinline__ void block_sum2(double& a, double& b, double (*red)[2], double* out2) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_xor(a, off, 64);
    b += __shfl_xor(b, off, 64);
  }
  if (lane == 0) { red[wv][0] = a; red[wv][1] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s0 = 0.0, s1 = 0.0;
    for (int w = 0; w < PNT / 64; ++w) { s0 += red[w][0]; s1 += red[w][1]; }
    out2[0] = s0;
    out2[1] = s1;
  }
  __syncthreads();
  a = out2[0];
  b = out2[1];
  __syncthreads();
}

__global__ __launch_bounds__(PNT) void focal_kernel(const float* __restrict__ pts, const float* __restrict__ conf, float4v* __restrict__ work,
                                                    float* __restrict__ focal_out, float* __restrict__ thr_out, int H, int W, float q,
                                                    float ppx, float ppy, int n_iter, float min_focal, float max_focal) {
  __shared__ uint32_t hist[2048];
  __shared__ int64_t sh_i64[2];
  __shared__ double red[PNT / 64][2];
  __shared__ double out2[2];
  __shared__ float sh_thr;
  const int64_t npix = (int64_t)H * W;
  const int64_t prob = blockIdx.x;
  const float* cf = conf + prob * npix;
  const float* pt = pts + prob * npix * 3;
  float4v* wk = work + prob * npix;

  // ---- threshold = torch.quantile(conf, q) (linear interpolation, fp32 rank and at::lerp; as in align_stats_kernel)
  const float rank = q * (float)(npix - 1);
  const float rlo = floorf(rank);
  const int64_t klo = (int64_t)rlo;
  const int64_t khi = (int64_t)ceilf(rank);
  const float vlo = fkey_inv(select_kth(cf, npix, klo, hist, sh_i64));
  float vhi = vlo;
  if (khi != klo) vhi = fkey_inv(select_kth(cf, npix, khi, hist, sh_i64));
  if (threadIdx.x == 0) {
    const float w = rank - rlo;
    const float d = vhi - vlo;
    sh_thr = (w < 0.5f) ? vlo + w * d : vhi - d * (1.0f - w);
  }
  __syncthreads();
  const float thr = sh_thr;
  if (threadIdx.x == 0 && thr_out) thr_out[prob] = thr;

  // ---- pass 0: per-point terms + the closed-form start  focal = mean(dot_xy_px) / mean(dot_xy_xy)  (post_process.py:121-128)
  double s_px = 0.0, s_xx = 0.0;
  double cnt = 0.0, dummy = 0.0;
  for (int64_t i = threadIdx.x; i < npix; i += PNT) {
    float4v rec = {0.f, 0.f, 0.f, 0.f};
    if (cf[i] >= thr) {
      const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
      const float X = pt[i * 3 + 0], Y = pt[i * 3 + 1], Z = pt[i * 3 + 2];
      float xz = __fdiv_rn(X, Z), yz = __fdiv_rn(Y, Z);
      if (!(fabsf(xz) <= 3.402823466e38f)) xz = 0.f;  // nan_to_num(posinf=0, neginf=0): nan, +inf, -inf -> 0
      if (!(fabsf(yz) <= 3.402823466e38f)) yz = 0.f;
      const float px = (float)x - ppx, py = (float)y - ppy;
      rec = float4v{px, py, xz, yz};
      s_px += (double)__fadd_rn(__fmul_rn(xz, px), __fmul_rn(yz, py));
      s_xx += (double)__fadd_rn(__fmul_rn(xz, xz), __fmul_rn(yz, yz));
      cnt += 1.0;
    }
    wk[i] = rec;
  }
  block_sum2(s_px, s_xx, red, out2);
  block_sum2(cnt, dummy, red, out2);
  const float focal_base = (float)((H > W ? H : W) / (2.0 * 0.57735026918962576451));  // max(H, W) / (2 tan(30 deg))
  if (cnt == 0.0) {  // post_process.py:102-105
    if (threadIdx.x == 0) focal_out[prob] = focal_base;
    return;
  }
  float focal = (float)(s_px / cnt) / (float)(s_xx / cnt);

  // ---- iteratively re-weighted least squares (post_process.py:131-136)
  for (int it = 0; it < n_iter; ++it) {
    double num = 0.0, den = 0.0;
    for (int64_t i = threadIdx.x; i < npix; i += PNT) {
      const float4v r = wk[i];
      const float dx = __fsub_rn(r[0], __fmul_rn(focal, r[2]));
      const float dy = __fsub_rn(r[1], __fmul_rn(focal, r[3]));
      const float dis = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
      const float w = __fdiv_rn(1.0f, fmaxf(dis, 1e-8f));
      const float dpx = __fadd_rn(__fmul_rn(r[2], r[0]), __fmul_rn(r[3], r[1]));
      const float dxx = __fadd_rn(__fmul_rn(r[2], r[2]), __fmul_rn(r[3], r[3]));
      num += (double)__fmul_rn(w, dpx);
      den += (double)__fmul_rn(w, dxx);
    }
    block_sum2(num, den, red, out2);
    focal = (float)num / (float)den;
  }
  if (threadIdx.x == 0) {
    const float lo = min_focal * focal_base, hi = max_focal * focal_base;
    focal_out[prob] = fminf(fmaxf(focal, lo), hi);  // post_process.py:140-142
  }
}

}  // namespace

extern "C" size_t f3r_align_workspace_bytes(int n_prob) { return (size_t)(n_prob > 0 ? n_prob : 0) * WS_PER * sizeof(double); }

extern "C" int f3r_align_local_to_global(const float* conf, const float* pts_local, const float* pts_global, const uint8_t* valid_mask,
                                         float* out, float* rts, float* thr_out, void* workspace, size_t ws_bytes, int n_prob,
                                         int64_t npix, float quantile, f3r_stream_t stream) {
  F3R_REQUIRE(conf && pts_local && pts_global && out && rts && workspace, "f3r_align_local_to_global: null pointer");
  F3R_REQUIRE(n_prob >= 0 && npix > 0, "f3r_align_local_to_global: bad sizes");
  F3R_REQUIRE(quantile >= 0.f && quantile <= 1.f, "f3r_align_local_to_global: quantile %f outside [0, 1]", (double)quantile);
  F3R_REQUIRE(ws_bytes >= f3r_align_workspace_bytes(n_prob) && (((uintptr_t)workspace) & 7) == 0, "f3r_align_local_to_global: workspace too small / misaligned");
  if (n_prob == 0) return F3R_OK;
  hipStream_t s = (hipStream_t)stream;
  double* ws = (double*)workspace;
  hipLaunchKernelGGL(align_stats_kernel, dim3(n_prob), dim3(PNT), 0, s, conf, pts_local, pts_global, valid_mask, ws, npix, quantile);
  hipLaunchKernelGGL(align_solve_kernel, dim3((n_prob + 63) / 64), dim3(64), 0, s, ws, rts, n_prob);
  const int64_t total = (int64_t)n_prob * npix;
  hipLaunchKernelGGL(align_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pts_local, rts, out, npix, total);
  if (thr_out) hipLaunchKernelGGL(align_thr_kernel, dim3((n_prob + 63) / 64), dim3(64), 0, s, ws, thr_out, n_prob);
  return f3r_check_launch("f3r_align_local_to_global");
}

extern "C" size_t f3r_focal_workspace_bytes(int n_views, int H, int W) {
  return (n_views > 0 && H > 0 && W > 0) ? (size_t)n_views * (size_t)H * (size_t)W * 16u : 0u;
}

extern "C" int f3r_estimate_focal(const float* pts3d, const float* conf, float* focal, float* thr_out, void* workspace, size_t ws_bytes,
                                  int n_views, int H, int W, float quantile, float ppx, float ppy, int n_iter, float min_focal,
                                  float max_focal, f3r_stream_t stream) {
  F3R_REQUIRE(pts3d && conf && focal && workspace, "f3r_estimate_focal: null pointer");
  F3R_REQUIRE(n_views >= 0 && H > 0 && W > 0 && n_iter >= 0, "f3r_estimate_focal: bad sizes");
  F3R_REQUIRE(quantile >= 0.f && quantile <= 1.f, "f3r_estimate_focal: quantile %f outside [0, 1]", (double)quantile);
  F3R_REQUIRE(ws_bytes >= f3r_focal_workspace_bytes(n_views, H, W) && (((uintptr_t)workspace) & 15) == 0, "f3r_estimate_focal: workspace too small / misaligned");
  if (n_views == 0) return F3R_OK;
  hipLaunchKernelGGL(focal_kernel, dim3(n_views), dim3(PNT), 0, (hipStream_t)stream, pts3d, conf, (float4v*)workspace, focal, thr_out, H, W,
                     quantile, ppx, ppy, n_iter, min_focal, max_focal);
  return f3r_check_launch("f3r_estimate_focal");
}
