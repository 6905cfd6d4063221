// Epilogues shared by the two MFMA GEMM kernels (f3r_gemm.hip: 128x128 tile, f3r_gemm256.hip: 256x256 tile).
//
// Both kernels hand over a wave's accumulators as 16x16 fp32 fragments of v_mfma_f32_16x16x32 (lane l holds D[4*(l>>4) + j][l & 15],
// j = 0..3) over a wave sub-tile of MF m-fragments x NF n-fragments whose origin is (m_base, n_base).  Fragment mf covers rows
// m_base + (mf / MG) * 128 + (mf % MG) * 16 .. + 15 and fragment nf columns n_base + (nf / NG) * 128 + (nf % NG) * 16 .. + 15, where
// (MG, NG) = fragments per contiguous group: the 128-tile kernel has one group each (MG = MF, NG = NF), the 256-tile kernel takes
// its 128 x 64 sub-tile as two row groups and two column groups, one from each HALF TILE (see f3r_gemm256.hip).  Layout struct L.
//   default roles  (weights = MFMA A operand): fragment (nf, mf) at acc[nf * MF + mf]; a lane owns 4 consecutive n of one row m
//                  -> every epilogue access (bias, residual, fp32 / lowp stores) is a 16 B / 8 B vector along n;
//   swapped roles  (activations = MFMA A operand; V third of the QKV epilogue): fragment (mf, nf) at acc[mf * NF + nf]; a lane owns 4
//                  consecutive tokens of one channel and writes V transposed with 8 B stores.
#pragma once
#include "f3r_common.h"

// PAIRED (256-tile kernel, every role but QKV): the kernel feeds the weight rows of a 32-column group to the two MFMA fragments of the
// group in the order  row(fragment f, MFMA row i) = (i / 4) * 8 + f * 4 + i % 4,  so lane group fg holds columns fg*8 .. fg*8+3 in
// fragment 0 and fg*8+4 .. fg*8+7 in fragment 1: EIGHT consecutive columns per lane and row -> 16-byte lowp stores / loads, 64 - 128 B
// contiguous per row and instruction instead of 32 B (the epilogue of a K = 1024 tile is bound by the number of partial-line stores).
// The lane's 4 columns of fragment nf are  n_base + col(nf) + fg * LW + 0..3  in both layouts.
template <int NF_, int MF_, int NG_, int MG_, bool PAIRED_ = false>
struct GemmFragLayout {
  static constexpr int NF = NF_, MF = MF_, NG = NG_, MG = MG_;
  static constexpr bool PAIRED = PAIRED_;
  static constexpr int LW = PAIRED_ ? 8 : 4;
  static constexpr int STEP = PAIRED_ ? 2 : 1;  // fragments whose lane columns are contiguous
  static_assert(!PAIRED_ || NG_ == 2, "paired columns: two fragments per 32-column group");
  static __device__ __forceinline__ int row(int mf) { return (mf / MG) * 128 + (mf % MG) * 16; }
  static __device__ __forceinline__ int col(int nf) { return (nf / NG) * 128 + (nf % NG) * (PAIRED_ ? 4 : 16); }
  static __device__ __forceinline__ int col_end() { return ((NF - 1) / NG) * 128 + NG * 16; }  // one past the sub-tile's last column
};

// Exact (erf) GELU, nn.GELU() default (blocks.py:84).  erfc(|z|) by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): one v_rcp,
// five FMAs, one v_exp -- libm's erff costs ~40 instructions per element and made the fc1 epilogue as long as its K loop.
// Branch-free in the sign so there is no cancellation for x < 0:  gelu(x) = x*Phi(x),  Phi(-|x|) = erfc(|x|/sqrt2)/2.
// Max abs deviation from fp64 GELU over [-10, 10]: 3.4e-7 (well below the 16-bit rounding of the stored activation).
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  poly = __builtin_fmaf(poly, t, 1.421413741f);
  poly = __builtin_fmaf(poly, t, -0.284496736f);
  poly = __builtin_fmaf(poly, t, 0.254829592f);
  const float u = poly * t * __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);  // erfc(z)
  const float h = 0.5f * x * u;
  return x >= 0.f ? x - h : h;
}

// The same function on a 4-vector, written with vector operations so that the multiplies and FMAs become v_pk_mul_f32 / v_pk_fma_f32
// (two elements per instruction): ~11 instructions per element instead of 16.  |x| enters only through the rational argument t, the
// exponent uses x*x, and  x >= 0 ? x - h : h  ==  max(x, 0) - |h|  (h has the sign of x).  Same polynomial, same 3.4e-7 bound.
// (fmaxf would cost a second v_max per element: the compiler has to quiet a possible signalling NaN first.)
__device__ __forceinline__ float4v gelu_erf4(float4v x) {
  float4v t, e;
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x[i]), 0.3275911f * 0.70710678118654752440f, 1.0f));
  float4v poly = t * (0.5f * 1.061405429f) + (0.5f * -1.453152027f);  // the 0.5 of h = 0.5 x erfc(.) folded into the coefficients
  poly = poly * t + (0.5f * 1.421413741f);
  poly = poly * t + (0.5f * -0.284496736f);
  poly = poly * t + (0.5f * 0.254829592f);
  const float4v arg = (x * x) * (-0.5f * 1.44269504088896340736f);
#pragma unroll
  for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(arg[i]);
  const float4v h = x * (poly * t * e);
  float4v r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __builtin_fmaf(x[i] + __builtin_fabsf(x[i]), 0.5f, -__builtin_fabsf(h[i]));  // max(x, 0) = (x + |x|) / 2, exact
  return r;
}

// 4 fp32 -> 4 lowp (hi) and, when lo != nullptr, the 4 lowp remainders v - float(hi): hi + lo carries ~2x the significand bits
template <class T>
__device__ __forceinline__ void store4_split(uint16_t* hi, uint16_t* lo, float4v v) {
  u32x2 o;
  o[0] = pack2<T>(v[0], v[1]);
  o[1] = pack2<T>(v[2], v[3]);
  *(u32x2*)hi = o;
  if (lo) {
    u32x2 r;
    r[0] = pack2<T>(v[0] - lo_f<T>(o[0]), v[1] - hi_f<T>(o[0]));
    r[1] = pack2<T>(v[2] - lo_f<T>(o[1]), v[3] - hi_f<T>(o[1]));
    *(u32x2*)lo = r;
  }
}

template <class T>
__device__ __forceinline__ float4v load4_lp(const uint16_t* ptr) {
  const u32x2 r = *(const u32x2*)ptr;
  return float4v{lo_f<T>(r[0]), hi_f<T>(r[0]), lo_f<T>(r[1]), hi_f<T>(r[1])};
}

// STEP consecutive fragments of a lane = 4 * STEP consecutive columns: one 8- or 16-byte access per plane
template <class T, int STEP>
__device__ __forceinline__ void storeN_split(char* hi, char* lo, const float4v* v) {
  if constexpr (STEP == 1) {
    store4_split<T>((uint16_t*)hi, (uint16_t*)lo, v[0]);
  } else {
    u32x4 o;
    o[0] = pack2<T>(v[0][0], v[0][1]); o[1] = pack2<T>(v[0][2], v[0][3]);
    o[2] = pack2<T>(v[1][0], v[1][1]); o[3] = pack2<T>(v[1][2], v[1][3]);
    *(u32x4*)hi = o;
    if (lo) {
      u32x4 r;
      r[0] = pack2<T>(v[0][0] - lo_f<T>(o[0]), v[0][1] - hi_f<T>(o[0])); r[1] = pack2<T>(v[0][2] - lo_f<T>(o[1]), v[0][3] - hi_f<T>(o[1]));
      r[2] = pack2<T>(v[1][0] - lo_f<T>(o[2]), v[1][1] - hi_f<T>(o[2])); r[3] = pack2<T>(v[1][2] - lo_f<T>(o[3]), v[1][3] - hi_f<T>(o[3]));
      *(u32x4*)lo = r;
    }
  }
}
template <class T, int STEP>
__device__ __forceinline__ void loadN_lp(const char* ptr, float4v* v) {
  if constexpr (STEP == 1) {
    v[0] = load4_lp<T>((const uint16_t*)ptr);
  } else {
    const u32x4 r = *(const u32x4*)ptr;
    v[0] = float4v{lo_f<T>(r[0]), hi_f<T>(r[0]), lo_f<T>(r[1]), hi_f<T>(r[1])};
    v[1] = float4v{lo_f<T>(r[2]), hi_f<T>(r[2]), lo_f<T>(r[3]), hi_f<T>(r[3])};
  }
}

// fp8 copies of 4 * STEP consecutive values for the next F3R_SPLIT_X3F8 convolution (f3r_gemm_args.out_f8): e4m3 of v clamped to +-448
// (v_cvt_pk_fp8_f32 does not saturate) at hi8, e4m3 of (v - float(fp16(v))) * 2^12, clamped, at lo8.  fp16 only.
__device__ __forceinline__ uint32_t f8x4(float a, float b, float c, float d) {
  auto cl = [](float x) { return fminf(fmaxf(x, -448.f), 448.f); };
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(a), cl(b), 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(c), cl(d), w, true);
  return (uint32_t)w;
}
__device__ __forceinline__ float f16_rest(float x) { return (x - (float)(_Float16)x) * 4096.f; }
template <int STEP>
__device__ __forceinline__ void storeN_f8(char* hi8, char* lo8, const float4v* v) {
  if constexpr (STEP == 1) {
    *(uint32_t*)hi8 = f8x4(v[0][0], v[0][1], v[0][2], v[0][3]);
    *(uint32_t*)lo8 = f8x4(f16_rest(v[0][0]), f16_rest(v[0][1]), f16_rest(v[0][2]), f16_rest(v[0][3]));
  } else {
    u32x2 h, l;
    h[0] = f8x4(v[0][0], v[0][1], v[0][2], v[0][3]);
    h[1] = f8x4(v[1][0], v[1][1], v[1][2], v[1][3]);
    l[0] = f8x4(f16_rest(v[0][0]), f16_rest(v[0][1]), f16_rest(v[0][2]), f16_rest(v[0][3]));
    l[1] = f8x4(f16_rest(v[1][0]), f16_rest(v[1][1]), f16_rest(v[1][2]), f16_rest(v[1][3]));
    *(u32x2*)hi8 = h;
    *(u32x2*)lo8 = l;
  }
}

// ------------------------------------------------------------------ additive terms of the GENERIC epilogue
// out = act(acc + bias) + rowadd + res_f32 + res_lp (+lo) + res_lp2 (+lo).  Addresses are clamped into the matrix instead of
// predicated so the loads are unconditional.  term(i, nf, value) receives every loaded 4-vector.
template <class T, int NF, int MB, class F>
__device__ __forceinline__ void gemm_additive_terms(const f3r_gemm_args& p, const int64_t (&mc)[MB], const int (&nbc)[NF], F&& term) {
  if (p.rowadd) {
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const float* row = p.rowadd + (mc[i] / p.rowadd_div) * (int64_t)p.N;
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) term(i, nf, *(const float4v*)(row + nbc[nf]));
    }
  }
  if (p.res_f32) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) term(i, nf, *(const float4v*)(p.res_f32 + mc[i] * p.ldr_f32 + nbc[nf]));
  }
  if (p.res_lp) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) term(i, nf, load4_lp<T>((const uint16_t*)p.res_lp + mc[i] * p.ldr_lp + nbc[nf]));
    if (p.res_lp_lo) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) term(i, nf, load4_lp<T>((const uint16_t*)p.res_lp_lo + mc[i] * p.ldr_lp + nbc[nf]));
    }
  }
  if (p.res_lp2) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) term(i, nf, load4_lp<T>((const uint16_t*)p.res_lp2 + mc[i] * p.ldr_lp2 + nbc[nf]));
    if (p.res_lp2_lo) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) term(i, nf, load4_lp<T>((const uint16_t*)p.res_lp2_lo + mc[i] * p.ldr_lp2 + nbc[nf]));
    }
  }
}

// Accumulator initialisation with the additive terms (256-tile kernel, act == NONE): the residual loads are issued at the START of
// the workgroup, land while the first K-tiles are still in flight, and go straight into the accumulator registers -- an epilogue
// that had to load them would expose one HBM latency per batch of fragments with nothing on the CU to cover it (one workgroup per
// CU), and has no registers to prefetch into (128 accumulators + 64 prefetch + operands > 256).  Reading x before and writing it
// after is safe in place: every element is read and written by the same lane.  Summation order differs from the reference's
// (x + (acc + b) vs (acc + b) + x) by fp32 rounding only.
// SRC is a COMPILE-TIME selection of the term pattern (a run-time chain of optional terms made hipcc spill 200-290 registers):
//   F3R_ADD_NONE zeros; F3R_ADD_RES_F32 the fp32 residual (x + attn(..), x + mlp(..)); F3R_ADD_ROWADD the image-id rows;
//   F3R_ADD_RES_LP res_lp [+ res_lp_lo] [+ res_lp2 [+ res_lp2_lo]] (skip connections of the DPT head).
enum { F3R_ADD_NONE = 0, F3R_ADD_RES_F32 = 1, F3R_ADD_ROWADD = 2, F3R_ADD_RES_LP = 3, F3R_ADD_UNSUPPORTED = -1 };

inline int gemm_additive_pattern(const f3r_gemm_args& a) {
  const int n_kinds = (a.rowadd != nullptr) + (a.res_f32 != nullptr) + (a.res_lp != nullptr || a.res_lp2 != nullptr);
  if (n_kinds == 0) return F3R_ADD_NONE;
  if (n_kinds > 1 || (a.res_lp2 && !a.res_lp)) return F3R_ADD_UNSUPPORTED;
  return a.rowadd ? F3R_ADD_ROWADD : (a.res_f32 ? F3R_ADD_RES_F32 : F3R_ADD_RES_LP);
}

template <class T, class L, int SRC, bool SWAP>
__device__ __forceinline__ void gemm_acc_init_additive(const f3r_gemm_args& p, float4v* acc, int64_t m_base, int n_base, int lane) {
  constexpr int NF = L::NF, MF = L::MF;
  const int fr = lane & 15, fg = lane >> 4;
  // The bias goes in through the accumulators as well (act(sum + b) with the sum started at b): the epilogue then has no load that
  // a store could be waiting behind.
  if (SWAP) {  // swapped roles: a lane owns 4 tokens of ONE channel n = fr
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int n = n_base + L::col(nf) + fr;
      const float b = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) acc[mf * NF + nf] = float4v{b, b, b, b};
    }
    return;
  }
  if (SRC != F3R_ADD_ROWADD && m_base + L::row(MF - 1) + 16 <= p.M && n_base + L::col_end() <= p.N) {
    // interior sub-tile: wave-uniform base + one 32-bit lane offset per tensor (see the interior epilogues below)
    float4v b4[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
      b4[nf] = p.bias ? *(const float4v*)((const char*)(p.bias + n_base + L::col(nf)) + (uint32_t)(fg * L::LW * 4)) : float4v{0.f, 0.f, 0.f, 0.f};
    if (SRC == F3R_ADD_NONE) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc[nf * MF + mf] = b4[nf];
      return;
    }
    if (SRC == F3R_ADD_RES_F32) {
      const uint32_t off = (uint32_t)(fr * (int)p.ldr_f32 + fg * L::LW) * 4u;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const char* const row = (const char*)p.res_f32 + ((m_base + L::row(mf)) * p.ldr_f32 + n_base) * 4;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[nf * MF + mf] = *(const float4v*)(row + L::col(nf) * 4 + off);
      }
    } else {
      const uint32_t off = (uint32_t)(fr * (int)p.ldr_lp + fg * L::LW) * 2u;
      const uint32_t off2 = (uint32_t)(fr * (int)p.ldr_lp2 + fg * L::LW) * 2u;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int64_t ro = ((m_base + L::row(mf)) * p.ldr_lp + n_base) * 2;
#pragma unroll
        for (int nf = 0; nf < NF; nf += L::STEP) {
          float4v t[L::STEP];
          loadN_lp<T, L::STEP>((const char*)p.res_lp + ro + L::col(nf) * 2 + off, t);
#pragma unroll
          for (int q = 0; q < L::STEP; ++q) acc[(nf + q) * MF + mf] = t[q];
        }
      }
      if (p.res_lp_lo) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const int64_t ro = ((m_base + L::row(mf)) * p.ldr_lp + n_base) * 2;
#pragma unroll
          for (int nf = 0; nf < NF; nf += L::STEP) {
            float4v t[L::STEP];
            loadN_lp<T, L::STEP>((const char*)p.res_lp_lo + ro + L::col(nf) * 2 + off, t);
#pragma unroll
            for (int q = 0; q < L::STEP; ++q) acc[(nf + q) * MF + mf] += t[q];
          }
        }
      }
      if (p.res_lp2) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const int64_t ro = ((m_base + L::row(mf)) * p.ldr_lp2 + n_base) * 2;
#pragma unroll
          for (int nf = 0; nf < NF; nf += L::STEP) {
            float4v t[L::STEP];
            loadN_lp<T, L::STEP>((const char*)p.res_lp2 + ro + L::col(nf) * 2 + off2, t);
#pragma unroll
            for (int q = 0; q < L::STEP; ++q) acc[(nf + q) * MF + mf] += t[q];
          }
        }
        if (p.res_lp2_lo) {
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) {
            const int64_t ro = ((m_base + L::row(mf)) * p.ldr_lp2 + n_base) * 2;
#pragma unroll
            for (int nf = 0; nf < NF; nf += L::STEP) {
            float4v t[L::STEP];
            loadN_lp<T, L::STEP>((const char*)p.res_lp2_lo + ro + L::col(nf) * 2 + off2, t);
#pragma unroll
            for (int q = 0; q < L::STEP; ++q) acc[(nf + q) * MF + mf] += t[q];
          }
          }
        }
      }
    }
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) acc[nf * MF + mf] += b4[nf];
    return;
  }
  int nbc[NF];
  float4v bias4[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int nb = n_base + L::col(nf) + fg * L::LW;
    nbc[nf] = nb < p.N ? nb : p.N - 4;
    bias4[nf] = p.bias ? *(const float4v*)(p.bias + nbc[nf]) : float4v{0.f, 0.f, 0.f, 0.f};
  }
  if (SRC == F3R_ADD_NONE) {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) acc[nf * MF + mf] = bias4[nf];
    return;
  }
  int64_t mc[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int64_t m = m_base + L::row(mf) + fr;
    mc[mf] = m < p.M ? m : p.M - 1;
  }
  if (SRC == F3R_ADD_RES_F32) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) acc[nf * MF + mf] = *(const float4v*)(p.res_f32 + mc[mf] * p.ldr_f32 + nbc[nf]);
  } else if (SRC == F3R_ADD_ROWADD) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const float* row = p.rowadd + (mc[mf] / p.rowadd_div) * (int64_t)p.N;
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) acc[nf * MF + mf] = *(const float4v*)(row + nbc[nf]);
    }
  } else {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) acc[nf * MF + mf] = load4_lp<T>((const uint16_t*)p.res_lp + mc[mf] * p.ldr_lp + nbc[nf]);
    if (p.res_lp_lo) {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[nf * MF + mf] += load4_lp<T>((const uint16_t*)p.res_lp_lo + mc[mf] * p.ldr_lp + nbc[nf]);
    }
    if (p.res_lp2) {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[nf * MF + mf] += load4_lp<T>((const uint16_t*)p.res_lp2 + mc[mf] * p.ldr_lp2 + nbc[nf]);
      if (p.res_lp2_lo) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) acc[nf * MF + mf] += load4_lp<T>((const uint16_t*)p.res_lp2_lo + mc[mf] * p.ldr_lp2 + nbc[nf]);
      }
    }
  }
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) acc[nf * MF + mf] += bias4[nf];
}

#define F3R_EPI_WAIT_LOADS() __builtin_amdgcn_s_waitcnt(0x0F70)  /* vmcnt(0) */

// ------------------------------------------------------------------ interior fast paths
// A wave sub-tile that lies completely inside the matrix (a wave-uniform test) needs no per-row predicate, and every address is
//     wave-uniform base (SGPR pair, scalar arithmetic)  +  ONE 32-bit lane offset shared by all fragments of a tensor,
// i.e. `global_store v_off, v_data, s[base:base+1] offset:imm`: 3 instructions per fragment instead of ~20.  The general bodies below
// (64-bit multiplies, compares and exec masking per fragment, 64-bit divisions per row in the QKV roles) made the epilogue a quarter of
// the time of a K = 1024 tile (s_memtime stamps, profiles/r02_gemm256_tile_stamps.jsonl); they remain for the edge tiles.
template <class L>
__device__ __forceinline__ bool gemm_wave_interior(const f3r_gemm_args& p, int64_t m_base, int n_base) {
  return m_base + L::row(L::MF - 1) + 16 <= p.M && n_base + L::col_end() <= p.N;
}

template <int V>
struct EpiC {
  static constexpr int value = V;
};

// GENERIC role, additive terms already in the accumulators
template <class T, class L>
__device__ __forceinline__ void gemm_epilogue_generic_interior(const f3r_gemm_args& p, const float4v* acc, int64_t m_base, int n_base, int lane) {
  constexpr int NF = L::NF, MF = L::MF;
  const int fr = lane & 15, fg = lane >> 4;
  const uint32_t off_f32 = (uint32_t)(fr * (int)p.ldo_f32 + fg * L::LW) * 4u;
  const uint32_t off_lp = (uint32_t)(fr * (int)p.ldo_lp + fg * L::LW) * 2u;
  const uint32_t off_f8 = (uint32_t)(fr * p.N * 2 + fg * L::LW);
  const bool relu = p.act == F3R_ACT_RELU;  // wave-uniform branch per fragment; GELU is a compile-time variant of the body
  // KIND: 0 out_lp only, 1 out_lp + out_lp_lo, 2 out_f32 only, 3 any combination (wave-uniform tests per fragment)
  auto run = [&](auto act_c, auto kind_c) {
    constexpr int ACT = decltype(act_c)::value, KIND = decltype(kind_c)::value;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int64_t mrow = m_base + L::row(mf);
      char* const bf = (KIND >= 2 && p.out_f32) ? (char*)p.out_f32 + (mrow * p.ldo_f32 + n_base) * 4 : nullptr;
      const int64_t lp_row = (mrow * p.ldo_lp + n_base) * 2;
      char* const bl = (KIND != 2 && p.out_lp) ? (char*)p.out_lp + lp_row : nullptr;
      char* const bll = ((KIND == 1 || KIND == 3) && p.out_lp_lo) ? (char*)p.out_lp_lo + lp_row : nullptr;
      char* const br = (KIND == 3 && p.out_relu) ? (char*)p.out_relu + lp_row : nullptr;
      char* const brl = (KIND == 3 && p.out_relu_lo) ? (char*)p.out_relu_lo + lp_row : nullptr;
      char* const b8 = (KIND == 3 && p.out_f8) ? (char*)p.out_f8 + (mrow * p.N * 2 + n_base) : nullptr;            // rows of [N hi8 | N lo8]
      char* const br8 = (KIND == 3 && p.out_relu_f8) ? (char*)p.out_relu_f8 + (mrow * p.N * 2 + n_base) : nullptr;
#pragma unroll
      for (int nf = 0; nf < NF; nf += L::STEP) {
        float4v v[L::STEP];
#pragma unroll
        for (int q = 0; q < L::STEP; ++q) {
          v[q] = acc[(nf + q) * MF + mf];
          if (ACT == F3R_ACT_GELU) v[q] = gelu_erf4(v[q]);
          else if (relu) v[q] = float4v{fmaxf(v[q][0], 0.f), fmaxf(v[q][1], 0.f), fmaxf(v[q][2], 0.f), fmaxf(v[q][3], 0.f)};
        }
        const int cb = L::col(nf);
        if (KIND >= 2 && bf) {
#pragma unroll
          for (int q = 0; q < L::STEP; ++q) *(float4v*)(bf + (cb + q * 4) * 4 + off_f32) = v[q];
        }
        if (KIND != 2 && bl) storeN_split<T, L::STEP>(bl + cb * 2 + off_lp, (KIND == 1 || (KIND == 3 && bll)) ? bll + cb * 2 + off_lp : nullptr, v);
        if (KIND == 3 && b8) storeN_f8<L::STEP>(b8 + cb + off_f8, b8 + cb + off_f8 + p.N, v);
        if (KIND == 3 && (br || br8)) {
          float4v r[L::STEP];
#pragma unroll
          for (int q = 0; q < L::STEP; ++q) r[q] = float4v{fmaxf(v[q][0], 0.f), fmaxf(v[q][1], 0.f), fmaxf(v[q][2], 0.f), fmaxf(v[q][3], 0.f)};
          if (br) storeN_split<T, L::STEP>(br + cb * 2 + off_lp, brl ? brl + cb * 2 + off_lp : nullptr, r);
          if (br8) storeN_f8<L::STEP>(br8 + cb + off_f8, br8 + cb + off_f8 + p.N, r);
        }
      }
    }
  };
  auto by_kind = [&](auto act_c) {
    const bool f32 = p.out_f32 != nullptr, lp = p.out_lp != nullptr, lo = p.out_lp_lo != nullptr, rl = p.out_relu != nullptr || p.out_f8 != nullptr || p.out_relu_f8 != nullptr;
    if (lp && !f32 && !lo && !rl) run(act_c, EpiC<0>{});
    else if (lp && lo && !f32 && !rl) run(act_c, EpiC<1>{});
    else if (f32 && !lp && !rl) run(act_c, EpiC<2>{});
    else run(act_c, EpiC<3>{});
  };
  if (p.act == F3R_ACT_GELU) by_kind(EpiC<F3R_ACT_GELU>{});
  else by_kind(EpiC<F3R_ACT_NONE>{});
}

// q / k parts of the QKV role (bias already in the accumulators); rows < 2^31 (validated on the host) so that token -> (sequence, y, x)
// is 32-bit unsigned arithmetic with wave-uniform divisors
template <class T, class L>
__device__ __forceinline__ void gemm_epilogue_qk_interior(const f3r_gemm_args& p, const float4v* acc, int64_t m_base, int n_base, int lane) {
  constexpr int NF = L::NF, MF = L::MF, MB = 4;
  const int fr = lane & 15, fg = lane >> 4;
  const int Dq = p.qkv_dq ? p.qkv_dq : p.N / 3;
  const int Dkv = (p.N - Dq) / 2;
  const int part = n_base < Dq ? 0 : 1;
  const int Dm = part == 0 ? Dq : Dkv;
  const int col0 = part == 0 ? 0 : Dq;
  char* const dst = (char*)(part == 0 ? p.q : p.k);
  const float qs = (part == 0 && p.q_scale != 0.f) ? p.q_scale : 1.f;
  const uint32_t off = (uint32_t)(fr * Dm + fg * L::LW) * 2u;
  const uint32_t seq = (uint32_t)p.seq_len, rw = (uint32_t)(p.rope_w > 0 ? p.rope_w : 1);
#pragma unroll
  for (int mb = 0; mb < MF; mb += MB) {
    float4v c[MB][NF / 2], sn[MB][NF / 2];
    if (p.rope_cos) {
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const uint32_t m = (uint32_t)(m_base + L::row(mb + i)) + (uint32_t)fr;
        const uint32_t pos = m % seq;
        const uint32_t py = pos / rw, px = pos - py * rw;
        const uint32_t grp = m / rw;
#pragma unroll
        for (int j = 0; j < NF / 2; ++j) {
          const int h = ((n_base + L::col(2 * j)) >> 5) & 1;
          const int64_t toff = p.rope_mode == 1 ? (int64_t)grp * 32 + h * 16 : (int64_t)(h == 0 ? py : px) * 16;
          c[i][j] = *(const float4v*)(p.rope_cos + toff + fg * 4);
          sn[i][j] = *(const float4v*)(p.rope_sin + toff + fg * 4);
        }
      }
      F3R_EPI_WAIT_LOADS();
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      char* const row = dst + ((m_base + L::row(mb + i)) * Dm + (n_base - col0)) * 2;
      float4v v[NF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) v[nf] = acc[nf * MF + mb + i];
      if (p.rope_cos) {
#pragma unroll
        for (int j = 0; j < NF / 2; ++j) {
          const float4v a = v[2 * j], b = v[2 * j + 1];
          v[2 * j] = a * c[i][j] - b * sn[i][j];
          v[2 * j + 1] = b * c[i][j] + a * sn[i][j];
        }
      }
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) store4_split<T>((uint16_t*)(row + L::col(nf) * 2 + off), nullptr, v[nf] * qs);
    }
  }
}

// ------------------------------------------------------------------ default roles: GENERIC / CONVT / the q and k thirds of QKV
// ADD: the additive terms are applied here (128-tile kernel: two workgroups per CU cover each other's epilogue latency); otherwise
// they are already in the accumulators (gemm_acc_init_additive).
// Every batch of loads is followed by an explicit s_waitcnt vmcnt(0) OUTSIDE the per-fragment predication: hipcc's wait-count pass
// loses track of a load waited for inside a conditionally executed fragment body and would wait vmcnt(0) -- i.e. for the previous
// fragment's STORES too -- at the top of every body.
template <class T, int EPI, class L, bool ADD>
__device__ __forceinline__ void gemm_epilogue_default(const f3r_gemm_args& p, const float4v* acc, int64_t m_base, int n_base, int lane) {
  constexpr int NF = L::NF, MF = L::MF;
  if constexpr (!ADD && EPI != F3R_EPI_CONVT) {
    if (gemm_wave_interior<L>(p, m_base, n_base) && p.M <= 0x7fffffffll) {
      if constexpr (EPI == F3R_EPI_GENERIC) gemm_epilogue_generic_interior<T, L>(p, acc, m_base, n_base, lane);
      else gemm_epilogue_qk_interior<T, L>(p, acc, m_base, n_base, lane);
      return;
    }
  }
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int MB = 4;  // fragment rows per batch
  static_assert(MF % MB == 0, "MF must be a multiple of the batch");
  int nb[NF], nbc[NF];   // this lane's 4 columns of fragment nf, and the same clamped into [0, N-4] for loads
  float4v bias4[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    nb[nf] = n_base + L::col(nf) + fg * L::LW;
    nbc[nf] = nb[nf] < p.N ? nb[nf] : p.N - 4;
    bias4[nf] = (ADD && p.bias) ? *(const float4v*)(p.bias + nbc[nf]) : float4v{0.f, 0.f, 0.f, 0.f};  // !ADD: already in acc
  }
  if (ADD) F3R_EPI_WAIT_LOADS();
  if (EPI == F3R_EPI_GENERIC || EPI == F3R_EPI_CONVT) {
#pragma unroll
    for (int mb = 0; mb < MF; mb += MB) {
      int64_t m[MB], mc[MB];
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        m[i] = m_base + L::row(mb + i) + fr;
        mc[i] = m[i] < p.M ? m[i] : p.M - 1;
      }
      float4v add[MB][NF];
      if (EPI == F3R_EPI_GENERIC && ADD) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) add[i][nf] = float4v{0.f, 0.f, 0.f, 0.f};
        gemm_additive_terms<T, NF, MB>(p, mc, nbc, [&](int i, int nf, float4v v) { add[i][nf] += v; });
        F3R_EPI_WAIT_LOADS();
      }
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        if (m[i] >= p.M) continue;
        int64_t ct_base = 0;
        if (EPI == F3R_EPI_CONVT) {
          const int hw = p.ct_h * p.ct_w;
          const int b = (int)(m[i] / hw);
          const int rem = (int)(m[i] % hw);
          const int y = rem / p.ct_w, x = rem % p.ct_w;
          ct_base = (((int64_t)b * p.ct_h * p.ct_s + (int64_t)y * p.ct_s) * ((int64_t)p.ct_w * p.ct_s) + (int64_t)x * p.ct_s);
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          if (nb[nf] >= p.N) continue;
          float4v v = acc[nf * MF + mb + i] + bias4[nf];
          if (p.act == F3R_ACT_GELU) {
            v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]);
          } else if (p.act == F3R_ACT_RELU) {
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
          }
          if (EPI == F3R_EPI_GENERIC) {
            if (ADD) v += add[i][nf];
            if (p.out_f32) *(float4v*)(p.out_f32 + m[i] * p.ldo_f32 + nb[nf]) = v;
            if (p.out_lp)
              store4_split<T>((uint16_t*)p.out_lp + m[i] * p.ldo_lp + nb[nf],
                              p.out_lp_lo ? (uint16_t*)p.out_lp_lo + m[i] * p.ldo_lp + nb[nf] : nullptr, v);
            if (p.out_f8) storeN_f8<1>((char*)p.out_f8 + m[i] * p.N * 2 + nb[nf], (char*)p.out_f8 + m[i] * p.N * 2 + p.N + nb[nf], &v);
            if (p.out_relu || p.out_relu_f8) {
              const float4v r = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
              if (p.out_relu)
                store4_split<T>((uint16_t*)p.out_relu + m[i] * p.ldo_lp + nb[nf],
                                p.out_relu_lo ? (uint16_t*)p.out_relu_lo + m[i] * p.ldo_lp + nb[nf] : nullptr, r);
              if (p.out_relu_f8) storeN_f8<1>((char*)p.out_relu_f8 + m[i] * p.N * 2 + nb[nf], (char*)p.out_relu_f8 + m[i] * p.N * 2 + p.N + nb[nf], &r);
            }
          } else {  // CONVT scatter (pixel shuffle): n = (dy*s + dx)*cout + co
            const int tap = nb[nf] / p.ct_cout;
            const int co = nb[nf] - tap * p.ct_cout;
            const int dy = tap / p.ct_s, dx = tap - dy * p.ct_s;
            const int64_t pix = ct_base + (int64_t)dy * ((int64_t)p.ct_w * p.ct_s) + dx;
            store4_split<T>((uint16_t*)p.out_lp + pix * p.ct_cout + co, p.out_lp_lo ? (uint16_t*)p.out_lp_lo + pix * p.ct_cout + co : nullptr, v);
          }
        }
      }
    }
  } else {  // ------------------------------------------------------------ QKV, q or k third
    // A column group of a wave (NG fragments = NG*16 columns, 32 or 64 wide and aligned) lies inside ONE head and inside whole 32-column
    // halves of it: RoPE pairs dim i with i + 16 inside a half, i.e. fragments (2j, 2j+1) of a lane.
    static_assert(EPI != F3R_EPI_QKV || (L::NG % 2 == 0 && NF % 2 == 0), "RoPE pairs need an even number of fragments per column group");
    // columns [0, Dq) are q, [Dq, Dq + Dkv) k (f3r_gemm_args.qkv_dq); wave-uniform: every column group of a wave lies in one part
    const int Dq = p.qkv_dq ? p.qkv_dq : p.N / 3;
    const int Dkv = (p.N - Dq) / 2;
    const int part = n_base < Dq ? 0 : 1;
    const int Dm = part == 0 ? Dq : Dkv;   // row stride of the destination
    const int col0 = part == 0 ? 0 : Dq;   // first GEMM column of the part
    uint16_t* dst = (uint16_t*)(part == 0 ? p.q : p.k);
    const float qs = (part == 0 && p.q_scale != 0.f) ? p.q_scale : 1.f;
#pragma unroll
    for (int mb = 0; mb < MF; mb += MB) {
      int64_t m[MB];
      float4v c[MB][NF / 2], sn[MB][NF / 2];
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        m[i] = m_base + L::row(mb + i) + fr;
        if (p.rope_cos) {
          // RoPE-2D (pos_embed.py:162-183): dims [0,32) of a head rotate by the row position y, [32,64) by the column position x
          const int64_t mc = m[i] < p.M ? m[i] : p.M - 1;
          const int pos = (int)(mc % p.seq_len);
          const int py = pos / p.rope_w, px = pos - py * p.rope_w;
          const int64_t grp = mc / p.rope_w;  // rope_mode 1: one angle set per row group (LlamaDecoder: per view)
#pragma unroll
          for (int j = 0; j < NF / 2; ++j) {
            const int h = ((n_base + L::col(2 * j)) >> 5) & 1;  // which 32-dim half of its head this fragment pair is (wave-uniform)
            const int64_t toff = p.rope_mode == 1 ? grp * 32 + h * 16 : (int64_t)(h == 0 ? py : px) * 16;
            c[i][j] = *(const float4v*)(p.rope_cos + toff + fg * 4);
            sn[i][j] = *(const float4v*)(p.rope_sin + toff + fg * 4);
          }
        }
      }
      if (p.rope_cos) F3R_EPI_WAIT_LOADS();
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        if (m[i] >= p.M) continue;
        float4v v[NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) v[nf] = acc[nf * MF + mb + i] + bias4[nf];
        if (p.rope_cos) {
#pragma unroll
          for (int j = 0; j < NF / 2; ++j) {
            const float4v a = v[2 * j], b = v[2 * j + 1];
            v[2 * j] = a * c[i][j] - b * sn[i][j];
            v[2 * j + 1] = b * c[i][j] + a * sn[i][j];
          }
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) store4_split<T>(dst + m[i] * (int64_t)Dm + (nb[nf] - col0), nullptr, v[nf] * qs);
      }
    }
  }
}

// ------------------------------------------------------------------ swapped roles: the V third of QKV -> vt[seq][d][token]
template <class T, class L, bool BIAS>
__device__ __forceinline__ void gemm_epilogue_vt(const f3r_gemm_args& p, const float4v* acc, int64_t m_base, int n_base, int lane) {
  constexpr int NF = L::NF, MF = L::MF;
  const int fr = lane & 15, fg = lane >> 4;
  const int Dq = p.qkv_dq ? p.qkv_dq : p.N / 3;
  const int Dm = (p.N - Dq) / 2;  // width of the k and of the v part
  uint16_t* vt = (uint16_t*)p.vt;
  const bool vec_ok = ((p.seq_len | p.ldvt) & 3) == 0;
  if (!BIAS && vec_ok && p.M <= 0x7fffffffll && gemm_wave_interior<L>(p, m_base, n_base)) {
    // interior sub-tile: token -> (sequence, position) once per fragment ROW in 32-bit arithmetic, address = A[mf] + B[nf]
    const uint32_t seq = (uint32_t)p.seq_len;
    int64_t A[MF], B[NF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const uint32_t mb = (uint32_t)(m_base + L::row(mf)) + (uint32_t)(fg * 4);
      const uint32_t sq = mb / seq, t = mb - sq * seq;
      A[mf] = (int64_t)sq * Dm * p.ldvt + t;
    }
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) B[nf] = (int64_t)(n_base + L::col(nf) + fr - Dq - Dm) * p.ldvt;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) store4_split<T>(vt + A[mf] + B[nf], nullptr, acc[mf * NF + nf]);
    return;
  }
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int n = n_base + L::col(nf) + fr;  // < N
    const int d = n - Dq - Dm;
    const float bb = (BIAS && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int64_t mb = m_base + L::row(mf) + fg * 4;
      if (mb >= p.M) continue;
      const float4v v = acc[mf * NF + nf] + bb;
      if (vec_ok) {  // seq_len % 4 == 0 -> the 4 tokens share a sequence; M % 4 == 0 follows
        const int64_t s = mb / p.seq_len, t = mb % p.seq_len;
        store4_split<T>(vt + (s * Dm + d) * p.ldvt + t, nullptr, v);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t m = mb + j;
          if (m < p.M) {
            const int64_t s = m / p.seq_len, t = m % p.seq_len;
            vt[(s * Dm + d) * p.ldvt + t] = to_lp<T>(v[j]);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------ fused tail of the DPT head (f3r_gemm_args.fin_w; 256 x 128 tile, N = 128)
// head[2]'s accumulators (bias already in them) -> act (head[3] ReLU) -> head[4]: 1x1 conv to 4 channels on the vector pipe in fp32 ->
// postprocess (dpt_block.py:375-381, heads/postprocess.py:16-64; the arithmetic of dpt_final_kernel in f3r_elem.hip).  The 128 channels of a
// pixel are spread over the 4 waves of a wave row (wn) and the 4 lane groups fg of a wave (8 consecutive channels per lane, PAIRED layout):
// every lane folds its 8 channels into 4 partial outputs per row (256 FMAs), two butterfly steps (lanes +-32, +-16) leave each lane with the
// fg-complete sums of 2 of its 8 fragment rows, the four wn partials meet in LDS in a fixed order (deterministic), and thread t < 256
// finishes row t of the tile.
template <class T, class L>
__device__ __forceinline__ void gemm_epilogue_fin(const f3r_gemm_args& p, const float4v* acc, int64_t m0_tile, int wm, int wn, int lane, int tid,
                                                  float* scratch /* LDS: [4 wn][256 rows][4] floats = 16 KiB */) {
  static_assert(L::NF == 2 && L::MF == 8 && L::PAIRED, "fin epilogue: the 256 x 128 tile's paired fragment layout");
  const int fr = lane & 15, fg = lane >> 4;
  const int c0 = wn * 32 + fg * 8;  // this lane's 8 channels (N == 128: one n-tile, n0 == 0)
  float w[4][8];
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const float4v w0 = *(const float4v*)(p.fin_w + o * p.N + c0), w1 = *(const float4v*)(p.fin_w + o * p.N + c0 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { w[o][j] = w0[j]; w[o][4 + j] = w1[j]; }
  }
  const bool relu = p.act == F3R_ACT_RELU;
  float s[8][4];
#pragma unroll
  for (int mf = 0; mf < 8; ++mf) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = acc[0 * 8 + mf][j]; v[4 + j] = acc[1 * 8 + mf][j]; }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float a = v[0] * w[o][0];
#pragma unroll
      for (int j = 1; j < 8; ++j) a = __builtin_fmaf(v[j], w[o][j], a);
      s[mf][o] = a;
    }
  }
  // butterfly over fg: lanes with fg >= 2 keep rows mf 4..7, the others 0..3; then fg odd keeps the upper pair of its four
  const bool up32 = fg >= 2, up16 = (fg & 1) != 0;
  float t[4][4], u[2][4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const float keep = up32 ? s[k + 4][o] : s[k][o], send = up32 ? s[k][o] : s[k + 4][o];
      t[k][o] = keep + __shfl_xor(send, 32, 64);
    }
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const float keep = up16 ? t[k + 2][o] : t[k][o], send = up16 ? t[k][o] : t[k + 2][o];
      u[k][o] = keep + __shfl_xor(send, 16, 64);
    }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int mf = (up32 ? 4 : 0) + (up16 ? 2 : 0) + k;
    const int row = wm * 64 + (mf >> 2) * 128 + (mf & 3) * 16 + fr;
    *(float4v*)(scratch + (wn * 256 + row) * 4) = float4v{u[k][0], u[k][1], u[k][2], u[k][3]};
  }
  __syncthreads();
  if (tid < 256) {
    const int64_t pix = m0_tile + tid;
    float4v a = *(const float4v*)(scratch + tid * 4);
#pragma unroll
    for (int q = 1; q < 4; ++q) a += *(const float4v*)(scratch + (q * 256 + tid) * 4);
    if (pix < p.M) {
      const float a0 = a[0] + p.fin_b[0], a1 = a[1] + p.fin_b[1], a2 = a[2] + p.fin_b[2], a3 = p.fin_n_out > 3 ? a[3] + p.fin_b[3] : 0.f;
      float sc = 1.f;  // reg_dense_depth (postprocess.py:27-51)
      if (p.fin_depth_mode != 1 /* F3R_DEPTH_LINEAR */) {
        const float d = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
        sc = (p.fin_depth_mode == 0 /* F3R_DEPTH_EXP */ ? expm1f(d) : d * d) / fmaxf(d, 1e-8f);
      }
      p.fin_pts[pix * 3 + 0] = a0 * sc;
      p.fin_pts[pix * 3 + 1] = a1 * sc;
      p.fin_pts[pix * 3 + 2] = a2 * sc;
      if (p.fin_conf)  // reg_dense_conf (:54-64)
        p.fin_conf[pix] = p.fin_conf_mode == 0 /* F3R_CONF_EXP */ ? p.fin_vmin + fminf(expf(a3), p.fin_vmax - p.fin_vmin)
                                                                  : (p.fin_vmax - p.fin_vmin) * (1.f / (1.f + expf(-a3))) + p.fin_vmin;
    }
  }
  __syncthreads();  // the scratch is K-tile buffer 2 again from the next tile's first phase on
}
