// HBM-bound kernels of the Fast3R path: patchify (im2col for the k=s=16 patch conv), LayerNorm/RMSNorm,
// bilinear x2 upsample (align_corners=True, NHWC), the fused last 1x1 conv + postprocess, fp32->lowp cast.
// All of them move 16 bytes per lane per access and keep statistics in fp32.
#include "f3r_common.h"

namespace {

// ------------------------------------------------------------------------------------------ patchify
// one thread = 8 consecutive dx of one (patch, c, dy) row: reads 2 x float4, writes 16 B
template <class T>
__global__ void patchify_kernel(const float* __restrict__ img, uint16_t* __restrict__ out, int B, int H, int W, int ps,
                                int64_t n_chunks) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_chunks) return;
  const int cpr = ps / 8;  // chunks per patch row
  const int K = 3 * ps * ps;
  const int cpp = K / 8;  // chunks per patch
  const int64_t patch = i / cpp;
  const int ck = (int)(i % cpp);
  const int c = ck / (ps * cpr);
  const int r2 = ck % (ps * cpr);
  const int dy = r2 / cpr, dx0 = (r2 % cpr) * 8;
  const int w = W / ps, h = H / ps;
  const int b = (int)(patch / ((int64_t)h * w));
  const int pr = (int)(patch % ((int64_t)h * w));
  const int py = pr / w, px = pr % w;
  const float* src = img + (((int64_t)b * 3 + c) * H + (py * ps + dy)) * W + px * ps + dx0;
  const float4v a = *(const float4v*)src;
  const float4v bq = *(const float4v*)(src + 4);
  u32x4 o;
  o[0] = pack2<T>(a[0], a[1]);
  o[1] = pack2<T>(a[2], a[3]);
  o[2] = pack2<T>(bq[0], bq[1]);
  o[3] = pack2<T>(bq[2], bq[3]);
  *(u32x4*)(out + patch * K + (int64_t)ck * 8) = o;
}

// Any patch size (DINOv2: 14): one thread = one PAIR of consecutive k of one patch row, scalar fp32 loads; k >= 3*ps*ps (the zero padding
// up to the row stride ld, a multiple of 8 so the GEMM can take the rows) is written as zeros.
template <class T>
__global__ void patchify_any_kernel(const float* __restrict__ img, uint16_t* __restrict__ out, int B, int H, int W, int ps, int ld, int64_t n_pairs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const int ppr = ld / 2;  // pairs per output row
  const int64_t patch = i / ppr;
  const int k0 = (int)(i % ppr) * 2;
  const int w = W / ps, h = H / ps;
  const int b = (int)(patch / ((int64_t)h * w));
  const int pr = (int)(patch % ((int64_t)h * w));
  const int py = pr / w, px = pr % w;
  float v[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = k0 + j;
    v[j] = 0.f;
    if (k < 3 * ps * ps) {
      const int c = k / (ps * ps), r = k % (ps * ps);
      const int dy = r / ps, dx = r % ps;
      v[j] = img[(((int64_t)b * 3 + c) * H + (py * ps + dy)) * W + px * ps + dx];
    }
  }
  *(uint32_t*)(out + patch * ld + k0) = pack2<T>(v[0], v[1]);
}

// ------------------------------------------------------------------------------------------ layernorm
// one wave per row; D % 4 == 0; float4 loads; two-pass (mean, then centred variance) from registers when
// D <= 2048 (8 float4 per lane), otherwise re-reads the row.
// ld_lp: row stride of out_lp in elements (D for the plain form).  F8: every row also gets an fp8 (e4m3) copy of the same values, clamped to
// +-448 (v_cvt_pk_fp8_f32 turns anything larger into NaN: tools/ubench/mfma_scale_probe.py), D elements behind its start (f3r_split W2F8).
template <class T, int MAXV, bool F8 = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, uint16_t* __restrict__ out_lp,
                                                        float* __restrict__ out_f32, int64_t rows, int D, float eps, int rms, int64_t ld_lp = 0) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * D;
  const int nv = D / 4;  // float4 per row
  float4v v[MAXV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      v[i] = *(const float4v*)(xr + idx * 4);
      sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    } else {
      v[i] = float4v{0.f, 0.f, 0.f, 0.f};
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  const float mean = rms ? 0.f : sum / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      const float4v d = v[i] - mean;
      sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
  const float rstd = rsqrtf(sq / (float)D + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      const float4v gm = *(const float4v*)(gamma + idx * 4);
      float4v y = (v[i] - mean) * rstd * gm;
      if (!rms && beta) y += *(const float4v*)(beta + idx * 4);
      if (out_f32) *(float4v*)(out_f32 + row * D + idx * 4) = y;
      if (out_lp) {
        u32x2 o;
        o[0] = pack2<T>(y[0], y[1]);
        o[1] = pack2<T>(y[2], y[3]);
        if (F8) {
          uint16_t* r16 = out_lp + row * ld_lp;
          *(u32x2*)(r16 + idx * 4) = o;
          float4v c;
#pragma unroll
          for (int t = 0; t < 4; ++t) c[t] = fminf(fmaxf(y[t], -448.f), 448.f);
          int w8 = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
          w8 = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], w8, true);
          *(int*)((uint8_t*)(r16 + D) + idx * 4) = w8;
        } else {
          *(u32x2*)(out_lp + row * D + idx * 4) = o;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ upsample x2 (align_corners)
// thread = 8 channels of one output pixel.  src = dst * (in-1)/(out_full-1), out_full = 2*in for F.interpolate(scale_factor=2,
// align_corners=True) -- or any other nominal size (f3r_interp_bilinear); the output may be cropped to (oh, ow).
template <class T>
__global__ void upsample2x_kernel(const uint16_t* __restrict__ in, const uint16_t* __restrict__ in_lo, uint16_t* __restrict__ out,
                                  uint16_t* __restrict__ out_lo, int B, int h, int w, int C, int oh, int ow, int full_h, int full_w,
                                  int64_t n_items, uint8_t* __restrict__ out_f8 = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items) return;
  const int cv = C / 8;
  const int c8 = (int)(i % cv);
  const int64_t pix = i / cv;
  const int ox = (int)(pix % ow);
  const int oy = (int)((pix / ow) % oh);
  const int b = (int)(pix / ((int64_t)ow * oh));
  // align_corners=True: src = dst * (in - 1) / (full_out - 1) (torch's area_pixel_compute_scale), full_out = the interpolation's own
  // output size (2 * in for the x2 upsamples, in * patch_size / 8 for the head's Interpolate); (oh, ow) <= it is the cropped extent
  const float sy = (full_h > 1) ? (float)(h - 1) / (float)(full_h - 1) : 0.f;
  const float sx = (full_w > 1) ? (float)(w - 1) / (float)(full_w - 1) : 0.f;
  const float fy = sy * (float)oy, fx = sx * (float)ox;
  int y0 = (int)fy, x0 = (int)fx;
  if (y0 > h - 1) y0 = h - 1;
  if (x0 > w - 1) x0 = w - 1;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const int64_t boff = (int64_t)b * h * w * C + c8 * 8;
  const int64_t o00 = boff + ((int64_t)y0 * w + x0) * C, o01 = boff + ((int64_t)y0 * w + x1) * C;
  const int64_t o10 = boff + ((int64_t)y1 * w + x0) * C, o11 = boff + ((int64_t)y1 * w + x1) * C;
  const u32x4 p00 = *(const u32x4*)(in + o00), p01 = *(const u32x4*)(in + o01), p10 = *(const u32x4*)(in + o10), p11 = *(const u32x4*)(in + o11);
  u32x4 q00 = {0u, 0u, 0u, 0u}, q01 = q00, q10 = q00, q11 = q00;  // low planes (a zero word is +0.0 in both 16-bit formats)
  if (in_lo) {
    q00 = *(const u32x4*)(in_lo + o00); q01 = *(const u32x4*)(in_lo + o01); q10 = *(const u32x4*)(in_lo + o10); q11 = *(const u32x4*)(in_lo + o11);
  }
  u32x4 o, ol;
  float y[8], yl[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // torch's upsample_bilinear2d: hy*(hx*p00 + lx*p01) + ly*(hx*p10 + lx*p11)
    const float a = hy * (hx * (lo_f<T>(p00[k]) + lo_f<T>(q00[k])) + lx * (lo_f<T>(p01[k]) + lo_f<T>(q01[k]))) +
                    ly * (hx * (lo_f<T>(p10[k]) + lo_f<T>(q10[k])) + lx * (lo_f<T>(p11[k]) + lo_f<T>(q11[k])));
    const float bq = hy * (hx * (hi_f<T>(p00[k]) + hi_f<T>(q00[k])) + lx * (hi_f<T>(p01[k]) + hi_f<T>(q01[k]))) +
                     ly * (hx * (hi_f<T>(p10[k]) + hi_f<T>(q10[k])) + lx * (hi_f<T>(p11[k]) + hi_f<T>(q11[k])));
    o[k] = pack2<T>(a, bq);
    y[2 * k] = a;
    y[2 * k + 1] = bq;
    yl[2 * k] = a - lo_f<T>(o[k]);
    yl[2 * k + 1] = bq - hi_f<T>(o[k]);
    ol[k] = pack2<T>(yl[2 * k], yl[2 * k + 1]);
  }
  *(u32x4*)(out + pix * C + c8 * 8) = o;
  if (out_lo) *(u32x4*)(out_lo + pix * C + c8 * 8) = ol;
  if (out_f8) {  // the fp8 planes of the next F3R_SPLIT_X3F8 convolution: per pixel [C hi8 | C lo8], lo8 = e4m3((y - fp16(y)) 2^12); both clamped to +-448
    auto cl = [](float x) { return fminf(fmaxf(x, -448.f), 448.f); };
    u32x2 h8, l8;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      int wv = __builtin_amdgcn_cvt_pk_fp8_f32(cl(y[4 * k]), cl(y[4 * k + 1]), 0, false);
      wv = __builtin_amdgcn_cvt_pk_fp8_f32(cl(y[4 * k + 2]), cl(y[4 * k + 3]), wv, true);
      h8[k] = (uint32_t)wv;
      int wl = __builtin_amdgcn_cvt_pk_fp8_f32(cl(yl[4 * k] * 4096.f), cl(yl[4 * k + 1] * 4096.f), 0, false);
      wl = __builtin_amdgcn_cvt_pk_fp8_f32(cl(yl[4 * k + 2] * 4096.f), cl(yl[4 * k + 3] * 4096.f), wl, true);
      l8[k] = (uint32_t)wl;
    }
    *(u32x2*)(out_f8 + pix * C * 2 + c8 * 8) = h8;
    *(u32x2*)(out_f8 + pix * C * 2 + C + c8 * 8) = l8;
  }
}

// ------------------------------------------------------------------------------------------ final 1x1 conv + postprocess
// thread = one pixel: 4 dot products over Cin (weights through LDS), then
//   d = |xyz|, pts = xyz / max(d, 1e-8) * expm1(d), conf = vmin + min(exp(c), vmax - vmin)   (postprocess.py:32-61)
template <class T>
__global__ __launch_bounds__(256) void dpt_final_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x_lo,
                                                        const float* __restrict__ w, const float* __restrict__ bias, int n_out,
                                                        float* __restrict__ pts, float* __restrict__ conf, int64_t npix, int Cin,
                                                        float vmin, float vmax, int depth_mode, int conf_mode) {
  extern __shared__ float wsh[];  // [4][Cin]; row 3 is zero when the head has no confidence channel (n_out == 3)
  for (int i = threadIdx.x; i < 4 * Cin; i += blockDim.x) wsh[i] = i < n_out * Cin ? w[i] : 0.f;
  __syncthreads();
  const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  float a0 = bias[0], a1 = bias[1], a2 = bias[2], a3 = n_out > 3 ? bias[3] : 0.f;
  const uint16_t* xp = x + pix * Cin;
  const uint16_t* xl = x_lo ? x_lo + pix * Cin : nullptr;
  for (int c = 0; c < Cin; c += 8) {
    const u32x4 v = *(const u32x4*)(xp + c);
    u32x4 vl = {0u, 0u, 0u, 0u};
    if (xl) vl = *(const u32x4*)(xl + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float f0 = lo_f<T>(v[k]) + lo_f<T>(vl[k]), f1 = hi_f<T>(v[k]) + hi_f<T>(vl[k]);
      const int cc = c + 2 * k;
      a0 = __builtin_fmaf(f0, wsh[cc], a0);            a0 = __builtin_fmaf(f1, wsh[cc + 1], a0);
      a1 = __builtin_fmaf(f0, wsh[Cin + cc], a1);      a1 = __builtin_fmaf(f1, wsh[Cin + cc + 1], a1);
      a2 = __builtin_fmaf(f0, wsh[2 * Cin + cc], a2);  a2 = __builtin_fmaf(f1, wsh[2 * Cin + cc + 1], a2);
      a3 = __builtin_fmaf(f0, wsh[3 * Cin + cc], a3);  a3 = __builtin_fmaf(f1, wsh[3 * Cin + cc + 1], a3);
    }
  }
  // reg_dense_depth (postprocess.py:27-51): 'linear' xyz; else direction xyz / max(d, 1e-8) times expm1(d) ('exp') or d^2 ('square')
  float sc = 1.f;
  if (depth_mode != F3R_DEPTH_LINEAR) {
    const float d = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
    sc = (depth_mode == F3R_DEPTH_EXP ? expm1f(d) : d * d) / fmaxf(d, 1e-8f);
  }
  pts[pix * 3 + 0] = a0 * sc;
  pts[pix * 3 + 1] = a1 * sc;
  pts[pix * 3 + 2] = a2 * sc;
  // reg_dense_conf (:54-64): 'exp' vmin + min(exp(x), vmax - vmin); 'sigmoid' (vmax - vmin) sigmoid(x) + vmin
  if (conf) conf[pix] = conf_mode == F3R_CONF_EXP ? vmin + fminf(expf(a3), vmax - vmin) : (vmax - vmin) * (1.f / (1.f + expf(-a3))) + vmin;
}

// Cin == 128 (the DPT head's last_dim): the one-pixel-per-lane walk above reads 16 B from 64 different 256-byte rows per load
// instruction (measured ~1 TB/s of the ~8 TB/s HBM roof).  Here a one-wave workgroup stages its 64 pixels by LDS-DMA -- every
// global_load_lds moves 1 KiB of CONTIGUOUS memory (4 pixels x 256 B) -- and each lane then reads its own pixel's row from LDS.  The
// 16-byte chunk c of pixel p is stored at chunk c ^ (p & 15) of the row (swizzle applied on the source address), so the 16 lanes of a
// ds_read_b128 group hit all 64 banks once.  16 KiB per plane; four to five workgroups per CU hide each other's DMA latency.
template <class T, bool HAS_LO>
__global__ __launch_bounds__(64) void dpt_final128_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x_lo,
                                                          const float* __restrict__ w, const float* __restrict__ bias, int n_out,
                                                          float* __restrict__ pts, float* __restrict__ conf, int64_t npix, float vmin,
                                                          float vmax, int depth_mode, int conf_mode) {
  constexpr int Cin = 128;
  __shared__ __attribute__((aligned(16))) uint16_t tile[(HAS_LO ? 2 : 1) * 64 * Cin];
  __shared__ float wsh[4 * Cin];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  const int lane = threadIdx.x;
  const int64_t p0 = (int64_t)blockIdx.x * 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int p = i * 4 + (lane >> 4);
    int64_t gp = p0 + p;
    if (gp >= npix) gp = npix - 1;
    const int64_t src = gp * Cin + (((lane & 15) ^ (p & 15)) << 3);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(x + src), (lds_ptr_t)(tile + i * 512), 16, 0, 0);
    if (HAS_LO) __builtin_amdgcn_global_load_lds((glb_ptr_t)(x_lo + src), (lds_ptr_t)(tile + 64 * Cin + i * 512), 16, 0, 0);
  }
  for (int i = lane; i < 4 * Cin; i += 64) wsh[i] = i < n_out * Cin ? w[i] : 0.f;
  __syncthreads();  // (the compiler drains vmcnt before the barrier: the tile has landed)
  float a0 = bias[0], a1 = bias[1], a2 = bias[2], a3 = n_out > 3 ? bias[3] : 0.f;
  const uint16_t* row = tile + lane * Cin;
#pragma unroll 4
  for (int c = 0; c < 16; ++c) {
    const int pc = (c ^ (lane & 15)) << 3;
    const u32x4 v = *(const u32x4*)(row + pc);
    u32x4 vl = {0u, 0u, 0u, 0u};
    if (HAS_LO) vl = *(const u32x4*)(row + 64 * Cin + pc);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float f0 = lo_f<T>(v[k]), f1 = hi_f<T>(v[k]);
      if (HAS_LO) { f0 += lo_f<T>(vl[k]); f1 += hi_f<T>(vl[k]); }
      const int cc = c * 8 + 2 * k;
      a0 = __builtin_fmaf(f0, wsh[cc], a0);            a0 = __builtin_fmaf(f1, wsh[cc + 1], a0);
      a1 = __builtin_fmaf(f0, wsh[Cin + cc], a1);      a1 = __builtin_fmaf(f1, wsh[Cin + cc + 1], a1);
      a2 = __builtin_fmaf(f0, wsh[2 * Cin + cc], a2);  a2 = __builtin_fmaf(f1, wsh[2 * Cin + cc + 1], a2);
      a3 = __builtin_fmaf(f0, wsh[3 * Cin + cc], a3);  a3 = __builtin_fmaf(f1, wsh[3 * Cin + cc + 1], a3);
    }
  }
  const int64_t pix = p0 + lane;
  if (pix >= npix) return;
  float sc = 1.f;  // postprocess exactly as in dpt_final_kernel
  if (depth_mode != F3R_DEPTH_LINEAR) {
    const float d = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
    sc = (depth_mode == F3R_DEPTH_EXP ? expm1f(d) : d * d) / fmaxf(d, 1e-8f);
  }
  pts[pix * 3 + 0] = a0 * sc;
  pts[pix * 3 + 1] = a1 * sc;
  pts[pix * 3 + 2] = a2 * sc;
  if (conf) conf[pix] = conf_mode == F3R_CONF_EXP ? vmin + fminf(expf(a3), vmax - vmin) : (vmax - vmin) * (1.f / (1.f + expf(-a3))) + vmin;
}

template <class T>
__global__ void cast_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, uint16_t* __restrict__ out_lo, int64_t n8) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const float4v a = *(const float4v*)(in + i * 8);
  const float4v b = *(const float4v*)(in + i * 8 + 4);
  u32x4 o;
  o[0] = pack2<T>(a[0], a[1]);
  o[1] = pack2<T>(a[2], a[3]);
  o[2] = pack2<T>(b[0], b[1]);
  o[3] = pack2<T>(b[2], b[3]);
  *(u32x4*)(out + i * 8) = o;
  if (out_lo) {  // lo = lowp(x - float(hi)): hi + lo carries ~2x the significand bits (f3r.h, f3r_split)
    u32x4 l;
    l[0] = pack2<T>(a[0] - lo_f<T>(o[0]), a[1] - hi_f<T>(o[0]));
    l[1] = pack2<T>(a[2] - lo_f<T>(o[1]), a[3] - hi_f<T>(o[1]));
    l[2] = pack2<T>(b[0] - lo_f<T>(o[2]), b[1] - hi_f<T>(o[2]));
    l[3] = pack2<T>(b[2] - lo_f<T>(o[3]), b[3] - hi_f<T>(o[3]));
    *(u32x4*)(out_lo + i * 8) = l;
  }
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
inline unsigned nblk(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }


// ------------------------------------------------------------------------------------------------------------------------
// Input pipeline: the resize of `load_images` (fast3r/dust3r/utils/image.py:68-74: PIL LANCZOS when
// shrinking, BICUBIC otherwise) and `ImgNorm` (:32).  Pillow's resize (src/libImaging/Resample.c) is integer work: per output
// coordinate a window [xmin, xmin + count) and 22-bit fixed-point weights (computed on the host in double precision, as Pillow does),
// an accumulator that starts at 1 << 21, an arithmetic shift by 22, saturation to 8 bits; horizontal pass into an 8-bit intermediate,
// then vertical pass.  One thread per output pixel (3 channels); bit-exact with PIL (tests/test_image.py).
__global__ void resample_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W, int axis, int out_size,
                                   const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ksize, int64_t n_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  // output geometry: axis 1 (horizontal): [H][out_size], axis 0 (vertical): [out_size][W]
  const int ow = axis == 1 ? out_size : W;
  const int oy = (int)(i / ow), ox = (int)(i - (int64_t)oy * ow);
  const int o = axis == 1 ? ox : oy;
  const int xmin = bounds[2 * o], cnt = bounds[2 * o + 1];
  const int32_t* k = kk + (int64_t)o * ksize;
  int32_t s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  if (axis == 1) {
    const uint8_t* p = in + ((int64_t)oy * W + xmin) * 3;
    for (int x = 0; x < cnt; ++x) {
      const int32_t w = k[x];
      s0 += (int32_t)p[3 * x] * w;
      s1 += (int32_t)p[3 * x + 1] * w;
      s2 += (int32_t)p[3 * x + 2] * w;
    }
  } else {
    const uint8_t* p = in + ((int64_t)xmin * W + ox) * 3;
    for (int y = 0; y < cnt; ++y) {
      const int32_t w = k[y];
      s0 += (int32_t)p[(int64_t)y * W * 3] * w;
      s1 += (int32_t)p[(int64_t)y * W * 3 + 1] * w;
      s2 += (int32_t)p[(int64_t)y * W * 3 + 2] * w;
    }
  }
  s0 >>= 22; s1 >>= 22; s2 >>= 22;  // arithmetic shift, as Resample.c's clip8
  uint8_t* q = out + i * 3;
  q[0] = (uint8_t)(s0 < 0 ? 0 : (s0 > 255 ? 255 : s0));
  q[1] = (uint8_t)(s1 < 0 ? 0 : (s1 > 255 ? 255 : s1));
  q[2] = (uint8_t)(s2 < 0 ? 0 : (s2 > 255 ? 255 : s2));
}

// crop box (x0, y0, w, h) of an [H][W][3] uint8 image -> fp32 [3][h][w], ((u / 255) - 0.5) / 0.5 in torchvision's operation order
__global__ void imgnorm_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int W, int x0, int y0, int w, int h) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)w * h) return;
  const int y = (int)(i / w), x = (int)(i - (int64_t)y * w);
  const uint8_t* p = in + ((int64_t)(y0 + y) * W + (x0 + x)) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = __fdiv_rn((float)p[c], 255.0f);
    out[(int64_t)c * w * h + i] = __fdiv_rn(__fsub_rn(v, 0.5f), 0.5f);
  }
}


// SwiGLU gate (llama.py:284): out[r][j] = silu(a) * b with a = ab[r][j], b = ab[r][hidden + j]; 8 columns per thread
template <class T>
__global__ void silu_mul_kernel(const uint16_t* __restrict__ ab, uint16_t* __restrict__ out, int64_t rows, int hidden, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int hv = hidden / 8;
  const int64_t r = i / hv;
  const int c = (int)(i - r * hv) * 8;
  const u32x4 a = *(const u32x4*)(ab + r * 2 * hidden + c);
  const u32x4 b = *(const u32x4*)(ab + r * 2 * hidden + hidden + c);
  u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a0 = lo_f<T>(a[k]), a1 = hi_f<T>(a[k]);
    const float s0 = a0 / (1.0f + __expf(-a0)), s1 = a1 / (1.0f + __expf(-a1));
    o[k] = pack2<T>(s0 * lo_f<T>(b[k]), s1 * hi_f<T>(b[k]));
  }
  *(u32x4*)(out + r * hidden + c) = o;
}

__global__ void rows_add_kernel(float* __restrict__ x, const float* __restrict__ vec, int D4, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4v* p = (float4v*)x + i;
  *p = *p + *((const float4v*)vec + (i % D4));
}

}  // namespace

#define F3R_DTYPE_OK(dt) F3R_REQUIRE((dt) == F3R_F16 || (dt) == F3R_BF16, "bad dtype %d", (dt))

extern "C" int f3r_patchify(const float* img, void* out, int batch, int H, int W, int ps, int ld_out, int dtype, f3r_stream_t stream) {
  F3R_REQUIRE(img && out && al16(img) && al16(out), "f3r_patchify: null/misaligned pointer");
  F3R_DTYPE_OK(dtype);
  F3R_REQUIRE(ps > 0 && H % ps == 0 && W % ps == 0 && batch >= 0, "f3r_patchify: H/W (%d,%d) not multiples of the patch size %d", H, W, ps);
  const int K = 3 * ps * ps;
  if (ld_out == 0) ld_out = K;
  F3R_REQUIRE(ld_out >= K && ld_out % 8 == 0, "f3r_patchify: row stride %d must be a multiple of 8 >= 3*ps*ps = %d", ld_out, K);
  const int64_t patches = (int64_t)batch * (H / ps) * (W / ps);
  if (patches == 0) return F3R_OK;
  hipStream_t s = (hipStream_t)stream;
  if (ps % 8 == 0 && ld_out == K) {  // 16 bytes per lane
    const int64_t n = patches * (K / 8);
    if (dtype == F3R_F16)
      hipLaunchKernelGGL(patchify_kernel<F16>, dim3(nblk(n, 256)), dim3(256), 0, s, img, (uint16_t*)out, batch, H, W, ps, n);
    else
      hipLaunchKernelGGL(patchify_kernel<BF16>, dim3(nblk(n, 256)), dim3(256), 0, s, img, (uint16_t*)out, batch, H, W, ps, n);
  } else {
    const int64_t n = patches * (ld_out / 2);
    if (dtype == F3R_F16)
      hipLaunchKernelGGL(patchify_any_kernel<F16>, dim3(nblk(n, 256)), dim3(256), 0, s, img, (uint16_t*)out, batch, H, W, ps, ld_out, n);
    else
      hipLaunchKernelGGL(patchify_any_kernel<BF16>, dim3(nblk(n, 256)), dim3(256), 0, s, img, (uint16_t*)out, batch, H, W, ps, ld_out, n);
  }
  return f3r_check_launch("f3r_patchify");
}

extern "C" int f3r_layernorm(const float* x, const float* gamma, const float* beta, void* out_lp, float* out_f32, int64_t rows, int D,
                             float eps, int rms, int dtype, f3r_stream_t stream) {
  F3R_REQUIRE(x && gamma && (out_lp || out_f32), "f3r_layernorm: null pointer");
  F3R_DTYPE_OK(dtype);
  F3R_REQUIRE(D > 0 && D % 4 == 0 && D <= 8192, "f3r_layernorm: D %d must be a multiple of 4, <= 8192", D);
  F3R_REQUIRE(al16(x) && al16(gamma) && (!beta || al16(beta)) && (!out_f32 || al16(out_f32)) && (!out_lp || (((uintptr_t)out_lp) & 7) == 0),
              "f3r_layernorm: misaligned pointer");
  if (rows <= 0) return F3R_OK;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = nblk(rows, 4);
#define LN_LAUNCH(TT, MV) \
  hipLaunchKernelGGL((layernorm_kernel<TT, MV>), dim3(grid), dim3(256), 0, s, x, gamma, beta, (uint16_t*)out_lp, out_f32, rows, D, eps, rms)
  const int nv = (D / 4 + 63) / 64;
  if (dtype == F3R_F16) {
    if (nv <= 4) LN_LAUNCH(F16, 4); else if (nv <= 8) LN_LAUNCH(F16, 8); else LN_LAUNCH(F16, 32);
  } else {
    if (nv <= 4) LN_LAUNCH(BF16, 4); else if (nv <= 8) LN_LAUNCH(BF16, 8); else LN_LAUNCH(BF16, 32);
  }
#undef LN_LAUNCH
  return f3r_check_launch("f3r_layernorm");
}

extern "C" int f3r_layernorm_f8(const float* x, const float* gamma, const float* beta, void* out_rows, int64_t ld_out, int64_t rows, int D, float eps, int rms,
                                f3r_stream_t stream) {
  F3R_REQUIRE(x && gamma && out_rows && al16(x) && al16(gamma) && al16(beta) && al16(out_rows), "f3r_layernorm_f8: null/misaligned pointer");
  F3R_REQUIRE(D > 0 && D % 8 == 0 && D <= 8192, "f3r_layernorm_f8: D %d must be a multiple of 8, <= 8192", D);
  F3R_REQUIRE(ld_out % 8 == 0 && ld_out * 2 >= (int64_t)D * 3, "f3r_layernorm_f8: ld_out %lld must be a multiple of 8 and >= 3 D / 2", (long long)ld_out);
  if (rows <= 0) return F3R_OK;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = nblk(rows, 4);
#define LN8_LAUNCH(MV) \
  hipLaunchKernelGGL((layernorm_kernel<F16, MV, true>), dim3(grid), dim3(256), 0, s, x, gamma, beta, (uint16_t*)out_rows, (float*)nullptr, rows, D, eps, rms, ld_out)
  const int nv = (D / 4 + 63) / 64;
  if (nv <= 4) LN8_LAUNCH(4); else if (nv <= 8) LN8_LAUNCH(8); else LN8_LAUNCH(32);
#undef LN8_LAUNCH
  return f3r_check_launch("f3r_layernorm_f8");
}

extern "C" int f3r_interp_bilinear(const void* in, const void* in_lo, void* out, void* out_lo, int batch, int h, int w, int C, int full_h,
                                   int full_w, int out_h, int out_w, int dtype, f3r_stream_t stream) {
  return f3r_interp_bilinear_f8(in, in_lo, out, out_lo, nullptr, batch, h, w, C, full_h, full_w, out_h, out_w, dtype, stream);
}

extern "C" int f3r_interp_bilinear_f8(const void* in, const void* in_lo, void* out, void* out_lo, void* out_f8, int batch, int h, int w, int C, int full_h,
                                      int full_w, int out_h, int out_w, int dtype, f3r_stream_t stream) {
  F3R_REQUIRE(in && out && al16(in) && al16(out) && al16(in_lo) && al16(out_lo), "f3r_interp_bilinear: null/misaligned pointer");
  F3R_DTYPE_OK(dtype);
  F3R_REQUIRE(!out_f8 || (dtype == F3R_F16 && (((uintptr_t)out_f8) & 7) == 0), "f3r_interp_bilinear_f8: the fp8 planes go with fp16 outputs (8-byte aligned)");
  F3R_REQUIRE(C > 0 && C % 8 == 0, "f3r_interp_bilinear: C %d must be a multiple of 8", C);
  F3R_REQUIRE(h > 0 && w > 0 && out_h > 0 && out_w > 0 && out_h <= full_h && out_w <= full_w, "f3r_interp_bilinear: bad sizes");
  const int64_t n = (int64_t)batch * out_h * out_w * (C / 8);
  if (n <= 0) return F3R_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == F3R_F16)
    hipLaunchKernelGGL(upsample2x_kernel<F16>, dim3(nblk(n, 256)), dim3(256), 0, s, (const uint16_t*)in, (const uint16_t*)in_lo, (uint16_t*)out,
                       (uint16_t*)out_lo, batch, h, w, C, out_h, out_w, full_h, full_w, n, (uint8_t*)out_f8);
  else
    hipLaunchKernelGGL(upsample2x_kernel<BF16>, dim3(nblk(n, 256)), dim3(256), 0, s, (const uint16_t*)in, (const uint16_t*)in_lo, (uint16_t*)out,
                       (uint16_t*)out_lo, batch, h, w, C, out_h, out_w, full_h, full_w, n, (uint8_t*)nullptr);
  return f3r_check_launch("f3r_interp_bilinear");
}

extern "C" int f3r_upsample2x(const void* in, const void* in_lo, void* out, void* out_lo, int batch, int h, int w, int C, int out_h,
                              int out_w, int dtype, f3r_stream_t stream) {
  return f3r_interp_bilinear(in, in_lo, out, out_lo, batch, h, w, C, 2 * h, 2 * w, out_h, out_w, dtype, stream);
}

extern "C" int f3r_dpt_final(const void* x, const void* x_lo, const float* w, const float* b, int n_out, float* pts3d, float* conf,
                             int64_t npix, int Cin, int depth_mode, int conf_mode, float conf_vmin, float conf_vmax, int dtype,
                             f3r_stream_t stream) {
  F3R_REQUIRE(x && w && b && pts3d && al16(x) && al16(x_lo), "f3r_dpt_final: null/misaligned pointer");
  F3R_DTYPE_OK(dtype);
  F3R_REQUIRE(Cin > 0 && Cin % 8 == 0 && Cin <= 2048, "f3r_dpt_final: Cin %d must be a multiple of 8, <= 2048", Cin);
  F3R_REQUIRE(n_out == 3 || n_out == 4, "f3r_dpt_final: n_out %d (3 = xyz, 4 = xyz + confidence)", n_out);
  F3R_REQUIRE(n_out == 4 || conf == nullptr, "f3r_dpt_final: a confidence output needs the 4-channel weight");
  F3R_REQUIRE(depth_mode >= F3R_DEPTH_EXP && depth_mode <= F3R_DEPTH_SQUARE, "f3r_dpt_final: bad depth_mode %d", depth_mode);
  F3R_REQUIRE(conf_mode == F3R_CONF_EXP || conf_mode == F3R_CONF_SIGMOID, "f3r_dpt_final: bad conf_mode %d", conf_mode);
  if (npix <= 0) return F3R_OK;
  hipStream_t s = (hipStream_t)stream;
  if (Cin == 128) {  // the DPT head: LDS-DMA staged, fully coalesced
    const dim3 grid(nblk(npix, 64)), block(64);
#define F3R_DPT128(TT, LO) hipLaunchKernelGGL((dpt_final128_kernel<TT, LO>), grid, block, 0, s, (const uint16_t*)x, (const uint16_t*)x_lo, w, b, n_out, \
                                              pts3d, conf, npix, conf_vmin, conf_vmax, depth_mode, conf_mode)
    if (dtype == F3R_F16) { if (x_lo) F3R_DPT128(F16, true); else F3R_DPT128(F16, false); }
    else { if (x_lo) F3R_DPT128(BF16, true); else F3R_DPT128(BF16, false); }
#undef F3R_DPT128
    return f3r_check_launch("f3r_dpt_final");
  }
  const size_t sh = (size_t)4 * Cin * sizeof(float);
  if (dtype == F3R_F16)
    hipLaunchKernelGGL(dpt_final_kernel<F16>, dim3(nblk(npix, 256)), dim3(256), sh, s, (const uint16_t*)x, (const uint16_t*)x_lo, w, b, n_out, pts3d,
                       conf, npix, Cin, conf_vmin, conf_vmax, depth_mode, conf_mode);
  else
    hipLaunchKernelGGL(dpt_final_kernel<BF16>, dim3(nblk(npix, 256)), dim3(256), sh, s, (const uint16_t*)x, (const uint16_t*)x_lo, w, b, n_out, pts3d,
                       conf, npix, Cin, conf_vmin, conf_vmax, depth_mode, conf_mode);
  return f3r_check_launch("f3r_dpt_final");
}

extern "C" int f3r_cast_f32_to_lp(const float* in, void* out, void* out_lo, int64_t n, int dtype, f3r_stream_t stream) {
  F3R_REQUIRE(in && out && al16(in) && al16(out) && al16(out_lo), "f3r_cast_f32_to_lp: null/misaligned pointer");
  F3R_DTYPE_OK(dtype);
  F3R_REQUIRE(n >= 0 && n % 8 == 0, "f3r_cast_f32_to_lp: n %lld must be a multiple of 8", (long long)n);
  if (n == 0) return F3R_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == F3R_F16)
    hipLaunchKernelGGL(cast_kernel<F16>, dim3(nblk(n / 8, 256)), dim3(256), 0, s, in, (uint16_t*)out, (uint16_t*)out_lo, n / 8);
  else
    hipLaunchKernelGGL(cast_kernel<BF16>, dim3(nblk(n / 8, 256)), dim3(256), 0, s, in, (uint16_t*)out, (uint16_t*)out_lo, n / 8);
  return f3r_check_launch("f3r_cast_f32_to_lp");
}

extern "C" int f3r_resample_u8(const uint8_t* in, uint8_t* out, int H, int W, int axis, int out_size, const int32_t* bounds,
                               const int32_t* kk, int ksize, f3r_stream_t stream) {
  F3R_REQUIRE(in && out && bounds && kk, "f3r_resample_u8: null pointer");
  F3R_REQUIRE(H > 0 && W > 0 && out_size > 0 && ksize > 0 && (axis == 0 || axis == 1), "f3r_resample_u8: bad sizes");
  const int64_t n = axis == 1 ? (int64_t)H * out_size : (int64_t)out_size * W;
  hipLaunchKernelGGL(resample_u8_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, H, W, axis, out_size, bounds, kk, ksize, n);
  return f3r_check_launch("f3r_resample_u8");
}

extern "C" int f3r_imgnorm_u8(const uint8_t* in, float* out, int H, int W, int x0, int y0, int w, int h, f3r_stream_t stream) {
  F3R_REQUIRE(in && out, "f3r_imgnorm_u8: null pointer");
  F3R_REQUIRE(H > 0 && W > 0 && w > 0 && h > 0 && x0 >= 0 && y0 >= 0 && x0 + w <= W && y0 + h <= H, "f3r_imgnorm_u8: crop box outside the image");
  hipLaunchKernelGGL(imgnorm_kernel, dim3(nblk((int64_t)w * h, 256)), dim3(256), 0, (hipStream_t)stream, in, out, W, x0, y0, w, h);
  return f3r_check_launch("f3r_imgnorm_u8");
}

extern "C" int f3r_silu_mul(const void* ab, void* out, int64_t rows, int hidden, int dtype, f3r_stream_t stream) {
  F3R_REQUIRE(ab && out && al16(ab) && al16(out), "f3r_silu_mul: null/misaligned pointer");
  F3R_DTYPE_OK(dtype);
  F3R_REQUIRE(rows >= 0 && hidden > 0 && hidden % 8 == 0, "f3r_silu_mul: hidden %d must be a positive multiple of 8", hidden);
  const int64_t n = rows * (hidden / 8);
  if (n == 0) return F3R_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == F3R_F16)
    hipLaunchKernelGGL(silu_mul_kernel<F16>, dim3(nblk(n, 256)), dim3(256), 0, s, (const uint16_t*)ab, (uint16_t*)out, rows, hidden, n);
  else
    hipLaunchKernelGGL(silu_mul_kernel<BF16>, dim3(nblk(n, 256)), dim3(256), 0, s, (const uint16_t*)ab, (uint16_t*)out, rows, hidden, n);
  return f3r_check_launch("f3r_silu_mul");
}

extern "C" int f3r_rows_add_f32(float* x, const float* vec, int64_t rows, int D, f3r_stream_t stream) {
  F3R_REQUIRE(x && vec && al16(x) && al16(vec), "f3r_rows_add_f32: null/misaligned pointer");
  F3R_REQUIRE(rows >= 0 && D > 0 && D % 4 == 0, "f3r_rows_add_f32: D %d must be a positive multiple of 4", D);
  const int64_t n = rows * (D / 4);
  if (n == 0) return F3R_OK;
  hipLaunchKernelGGL(rows_add_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, x, vec, D / 4, n);
  return f3r_check_launch("f3r_rows_add_f32");
}
