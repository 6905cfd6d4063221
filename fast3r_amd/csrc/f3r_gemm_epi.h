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

template <int NF_, int MF_, int NG_, int MG_>
struct GemmFragLayout {
  static constexpr int NF = NF_, MF = MF_, NG = NG_, MG = MG_;
  static __device__ __forceinline__ int row(int mf) { return (mf / MG) * 128 + (mf % MG) * 16; }
  static __device__ __forceinline__ int col(int nf) { return (nf / NG) * 128 + (nf % NG) * 16; }
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
  int nbc[NF];
  float4v bias4[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int nb = n_base + L::col(nf) + fg * 4;
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

// ------------------------------------------------------------------ default roles: GENERIC / CONVT / the q and k thirds of QKV
// ADD: the additive terms are applied here (128-tile kernel: two workgroups per CU cover each other's epilogue latency); otherwise
// they are already in the accumulators (gemm_acc_init_additive).
// Every batch of loads is followed by an explicit s_waitcnt vmcnt(0) OUTSIDE the per-fragment predication: hipcc's wait-count pass
// loses track of a load waited for inside a conditionally executed fragment body and would wait vmcnt(0) -- i.e. for the previous
// fragment's STORES too -- at the top of every body.
#define F3R_EPI_WAIT_LOADS() __builtin_amdgcn_s_waitcnt(0x0F70)
template <class T, int EPI, class L, bool ADD>
__device__ __forceinline__ void gemm_epilogue_default(const f3r_gemm_args& p, const float4v* acc, int64_t m_base, int n_base, int lane) {
  constexpr int NF = L::NF, MF = L::MF;
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int MB = 4;  // fragment rows per batch
  static_assert(MF % MB == 0, "MF must be a multiple of the batch");
  int nb[NF], nbc[NF];   // this lane's 4 columns of fragment nf, and the same clamped into [0, N-4] for loads
  float4v bias4[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    nb[nf] = n_base + L::col(nf) + fg * 4;
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
            if (p.out_relu) {
              const float4v r = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
              store4_split<T>((uint16_t*)p.out_relu + m[i] * p.ldo_lp + nb[nf],
                              p.out_relu_lo ? (uint16_t*)p.out_relu_lo + m[i] * p.ldo_lp + nb[nf] : nullptr, r);
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
