// f3r_attn_fwd for head dimensions other than 64 (f3r_attn_args.head_dim: a multiple of 16 up to 128): the reference's Attention takes any
// dim // num_heads (croco/models/blocks.py:113-143) and its own scaling ablation runs a head_dim-80 fusion decoder
// (configs/experiment/model_scaling/model_scaling_huge.yaml:13-15: 1280 / 16 heads).  head_dim 64 -- every released checkpoint -- stays on
// the tuned kernels (f3r_attn.hip, csrc/asm/attn_gen.py); this one is the same math laid out for any width, not tuned further:
//   * workgroup = 4 waves x 32 queries; 64-key tiles of K [64][HD] and V^T [HD][64] staged through LDS (padded rows, 16-byte vector
//     loads / stores, one tile in flight);
//   * swapped Q K^T on v_mfma_f32_32x32x16 with the K rows fed through pi = swap(bit 2, bit 3) so that the score accumulators are the
//     B operand of P V after a pack (as in f3r_attn.hip); HD / 16 k-steps, ceil(HD / 32) output blocks of O^T (the rows past HD of the
//     last block multiply zeros that sit in LDS and are never stored);
//   * classic online softmax in fp32 (per-tile row max, exp2 units), partial tiles masked, K/V segments, carried (m, l, O) state in the
//     layout of f3r_attn.hip generalised to HD columns per head, grouped-query heads.  Causal masking is only built for head_dim 64.
#include "f3r_common.h"

namespace {

constexpr int GQ = 128;  // queries per workgroup (4 waves x 32)

template <class T, int HD>
__global__ __launch_bounds__(256) void attn_generic_kernel(const f3r_attn_args p) {
  constexpr int KS = HD / 16;               // k-steps of Q K^T
  constexpr int DB = (HD + 31) / 32;        // 32-row blocks of O^T
  constexpr int KLD = HD + 8;               // K tile row stride (elements): 16-byte aligned, de-phases the banks
  constexpr int VLD = 64 + 8;               // V^T tile row stride
  __shared__ __attribute__((aligned(16))) uint16_t kt[64 * KLD];
  __shared__ __attribute__((aligned(16))) uint16_t vt[DB * 32 * VLD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lq = lane & 31, g = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int kv_head = p.kv_group > 1 ? head / p.kv_group : head;
  const int64_t q0 = (int64_t)blockIdx.x * GQ + wid * 32;
  int64_t qrow = q0 + lq;
  const bool q_ok = qrow < p.tq;
  if (!q_ok) qrow = p.tq - 1;
  const float cq = p.q_prescaled ? 1.0f : p.scale * 1.44269504088896340736f;

  // rows HD .. DB*32-1 of the V^T image are never loaded: zero them once
  for (int i = tid; i < (DB * 32 - HD) * VLD; i += 256) vt[HD * VLD + i] = 0;

  typename T::vec8 qf[KS];
  {
    const uint16_t* Qg = (const uint16_t*)p.q + (int64_t)b * p.q_batch_stride + qrow * p.ldq + (int64_t)head * HD;
#pragma unroll
    for (int ds = 0; ds < KS; ++ds) {
      u32x4 raw = *(const u32x4*)(Qg + ds * 16 + g * 8);
      if (!p.q_prescaled) {
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = pack2<T>(lo_f<T>(raw[j]) * cq, hi_f<T>(raw[j]) * cq);
      }
      qf[ds] = as_vec8<T>(raw);
    }
  }
  float16v o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int64_t srow = (int64_t)b * p.tq + qrow;
  if (p.state_in) {
    const float* so = p.st_o + srow * ((int64_t)p.n_heads * HD) + (int64_t)head * HD;
    const float* sm = p.st_ml + (srow * p.n_heads + head) * 4;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int d = db * 32 + 8 * rq + 4 * g;
        if (d < HD) {
          const float4v v = *(const float4v*)(so + d);
          o[db][rq * 4 + 0] = v[0]; o[db][rq * 4 + 1] = v[1]; o[db][rq * 4 + 2] = v[2]; o[db][rq * 4 + 3] = v[3];
        }
      }
    m_run = sm[0];
    l_run = sm[1 + g];
  }
  const int krow_pi = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);

  for (int sg = 0; sg < p.n_seg; ++sg) {
    const int64_t n_keys = p.seg_len[sg];
    if (n_keys <= 0) continue;
    const uint16_t* Kg = (const uint16_t*)p.k_seg[sg] + (int64_t)b * p.k_batch_stride[sg] + (int64_t)kv_head * HD;
    const uint16_t* Vg = (const uint16_t*)p.vt_seg[sg] + (int64_t)b * p.vt_batch_stride[sg] + (int64_t)kv_head * HD * p.ldvt[sg];
    for (int64_t k0 = 0; k0 < n_keys; k0 += 64) {
      const int valid = (int)(n_keys - k0 < 64 ? n_keys - k0 : 64);
      __syncthreads();  // the previous tile is consumed
      for (int c = tid; c < 64 * (HD / 8); c += 256) {  // K tile: 64 rows x HD/8 chunks of 16 bytes
        const int row = c / (HD / 8), ch = c % (HD / 8);
        const int krow = row < valid ? row : valid - 1;
        *(u32x4*)(kt + row * KLD + ch * 8) = *(const u32x4*)(Kg + (k0 + krow) * p.ldk + ch * 8);
      }
      for (int c = tid; c < HD * 8; c += 256) {  // V^T tile: HD rows x 8 chunks (the rows are zero padded to ldvt, a multiple of 64)
        const int row = c >> 3, ch = c & 7;
        *(u32x4*)(vt + row * VLD + ch * 8) = *(const u32x4*)(Vg + (int64_t)row * p.ldvt[sg] + k0 + ch * 8);
      }
      __syncthreads();
      // ---- S^T = K Q^T (exp2 units)
      float16v s[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
        for (int ds = 0; ds < KS; ++ds) {
          const typename T::vec8 a = as_vec8<T>(*(const u32x4*)(kt + (kb * 32 + krow_pi) * KLD + ds * 16 + g * 8));
          s[kb] = T::mfma32(a, qf[ds], s[kb]);
        }
      }
      // register r of block kb is key kb*32 + 16*(r>>3) + 8*g + (r&7)
      float tmax = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kc = kb * 32 + 16 * (r >> 3) + 8 * g + (r & 7);
          if (kc >= valid) s[kb][r] = -INFINITY;
          tmax = fmaxf(tmax, s[kb][r]);
        }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m_run, tmax);  // finite: every tile holds a valid key
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
      typename T::vec8 pf[4];
      float psum = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        u32x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p0 = __builtin_amdgcn_exp2f(s[ks >> 1][(ks & 1) * 8 + 2 * j] - m_new);
          const float p1 = __builtin_amdgcn_exp2f(s[ks >> 1][(ks & 1) * 8 + 2 * j + 1] - m_new);
          pk[j] = pack2<T>(p0, p1);
          psum += lo_f<T>(pk[j]) + hi_f<T>(pk[j]);  // the sum of what P V multiplies
        }
        pf[ks] = as_vec8<T>(pk);
      }
      l_run += psum;
      // ---- O^T += V^T P^T
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const typename T::vec8 a = as_vec8<T>(*(const u32x4*)(vt + (db * 32 + lq) * VLD + ks * 16 + g * 8));
          o[db] = T::mfma32(a, pf[ks], o[db]);
        }
    }
  }
  if (!q_ok) return;
  if (p.state_out) {
    float* so = p.st_o + srow * ((int64_t)p.n_heads * HD) + (int64_t)head * HD;
    float* sm = p.st_ml + (srow * p.n_heads + head) * 4;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int d = db * 32 + 8 * rq + 4 * g;
        if (d < HD) {
          float4v v = {o[db][rq * 4 + 0], o[db][rq * 4 + 1], o[db][rq * 4 + 2], o[db][rq * 4 + 3]};
          *(float4v*)(so + d) = v;
        }
      }
    if (g == 0) sm[0] = m_run;
    sm[1 + g] = l_run;
    return;
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  uint16_t* Og = (uint16_t*)p.o + (int64_t)b * p.o_batch_stride + qrow * p.ldo + (int64_t)head * HD;
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int d = db * 32 + 8 * rq + 4 * g;
      if (d < HD) {
        u32x2 w;
        w[0] = pack2<T>(o[db][rq * 4 + 0] * inv, o[db][rq * 4 + 1] * inv);
        w[1] = pack2<T>(o[db][rq * 4 + 2] * inv, o[db][rq * 4 + 3] * inv);
        *(u32x2*)(Og + d) = w;
      }
    }
}

template <class T, int HD>
int launch_hd(const f3r_attn_args& a, hipStream_t s) {
  const int64_t qblocks = (a.tq + GQ - 1) / GQ;
  F3R_REQUIRE(qblocks < (1ll << 31) && a.n_heads < 65536 && a.batch < 65536, "f3r_attn_fwd: grid too large");
  hipLaunchKernelGGL((attn_generic_kernel<T, HD>), dim3((unsigned)qblocks, (unsigned)a.n_heads, (unsigned)a.batch), dim3(256), 0, s, a);
  return f3r_check_launch("f3r_attn_fwd(generic head_dim)");
}

template <class T>
int launch_t(const f3r_attn_args& a, hipStream_t s) {
  switch (a.head_dim) {
    case 16: return launch_hd<T, 16>(a, s);
    case 32: return launch_hd<T, 32>(a, s);
    case 48: return launch_hd<T, 48>(a, s);
    case 80: return launch_hd<T, 80>(a, s);
    case 96: return launch_hd<T, 96>(a, s);
    case 112: return launch_hd<T, 112>(a, s);
    case 128: return launch_hd<T, 128>(a, s);
    default: break;
  }
  f3r_set_error("f3r_attn_fwd: head_dim %d is not built (16, 32, 48, 64, 80, 96, 112, 128)", a.head_dim);
  return F3R_ERR_UNSUPPORTED;
}

}  // namespace

int f3r_attn_generic_launch(const f3r_attn_args& a, hipStream_t s) {
  if (a.causal) {
    f3r_set_error("f3r_attn_fwd: causal attention is only built for head_dim 64");
    return F3R_ERR_UNSUPPORTED;
  }
  return a.dtype == F3R_F16 ? launch_t<F16>(a, s) : launch_t<BF16>(a, s);
}
