// f3r_attn_fwd for head dimensions other than 64 (f3r_attn_args.head_dim: a multiple of 16 up to 128): the reference's Attention takes any
// dim // num_heads (croco/models/blocks.py:113-143) and its own scaling ablation runs a head_dim-80 fusion decoder
// (configs/experiment/model_scaling/model_scaling_huge.yaml:13-15: 1280 / 16 heads).  head_dim 64 -- every released checkpoint -- stays on
// the tuned kernels (f3r_attn.hip, csrc/asm/attn_gen.py); this one is the same math laid out for any width:
//   * workgroup = 4 waves x 64 queries (two 32-query blocks per wave: every K / V^T fragment read from LDS feeds two MFMAs; one block
//     per wave above head_dim 96, where two sets of accumulators no longer fit the register file); 64-key
//     tiles of K [64][HD] and V^T [HD][64] in LDS (padded rows, 16-byte vector accesses), the NEXT tile's global loads are issued into
//     registers before the current tile is computed and stored to LDS after it (one LDS image, loads hidden behind the math);
//   * swapped Q K^T on v_mfma_f32_32x32x16 with the K rows fed through pi = swap(bit 2, bit 3) so that the score accumulators are the
//     B operand of P V after a pack (as in f3r_attn.hip); HD / 16 k-steps, ceil(HD / 32) output blocks of O^T (the rows past HD of the
//     last block multiply zeros that sit in LDS and are never stored);
//   * online softmax in fp32, exp2 units: the running reference m enters the scores through the C operand of the first Q K^T k-step, a
//     per-tile row maximum moves it (and rescales O, l, the tile's scores) only on tiles where some row exceeds it, row sums by v_dot2c on
//     the packed probabilities; partial tiles masked, K/V segments, carried (m, l, O) state in the layout of f3r_attn.hip generalised to HD columns per head,
//     grouped-query heads.  Causal masking is only built for head_dim 64.
#include "f3r_common.h"

namespace {

#ifndef F3R_GENERIC_QB2_MAX_HD
#define F3R_GENERIC_QB2_MAX_HD 96  // two query blocks per wave up to this head_dim, one above (register budget: O alone is QB * ceil(HD/32) * 16)
#endif

template <class T, int HD, int QB>  // QB = 32-query blocks per wave
__global__ __launch_bounds__(256) void attn_generic_kernel(const f3r_attn_args p) {
  constexpr int GQ = 4 * 32 * QB;           // queries per workgroup
  constexpr int KS = HD / 16;               // k-steps of Q K^T
  constexpr int DB = (HD + 31) / 32;        // 32-row blocks of O^T
  constexpr int KLD = HD + 8;               // K tile row stride (elements): 16-byte aligned, de-phases the banks
  constexpr int VLD = 64 + 8;               // V^T tile row stride
  constexpr int NCH = 64 * (HD / 8);        // 16-byte chunks of a K tile (64 rows x HD / 8) = of a V^T tile (HD rows x 8)
  constexpr int NLD = (NCH + 255) / 256;    // chunks of each tile a thread stages
  __shared__ __attribute__((aligned(16))) uint16_t kt[64 * KLD];
  __shared__ __attribute__((aligned(16))) uint16_t vt[DB * 32 * VLD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lq = lane & 31, g = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int kv_head = p.kv_group > 1 ? head / p.kv_group : head;
  const int64_t q0 = (int64_t)blockIdx.x * GQ + wid * (32 * QB);
  const float cq = p.q_prescaled ? 1.0f : p.scale * 1.44269504088896340736f;

  // rows HD .. DB*32-1 of the V^T image are never loaded: zero them once
  for (int i = tid; i < (DB * 32 - HD) * VLD; i += 256) vt[HD * VLD + i] = 0;

  int64_t qrow[QB];
  bool q_ok[QB];
  typename T::vec8 qf[QB][KS];
  float16v o[QB][DB];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    qrow[qb] = q0 + qb * 32 + lq;
    q_ok[qb] = qrow[qb] < p.tq;
    if (!q_ok[qb]) qrow[qb] = p.tq - 1;
    const uint16_t* Qg = (const uint16_t*)p.q + (int64_t)b * p.q_batch_stride + qrow[qb] * p.ldq + (int64_t)head * HD;
#pragma unroll
    for (int ds = 0; ds < KS; ++ds) {
      u32x4 raw = *(const u32x4*)(Qg + ds * 16 + g * 8);
      if (!p.q_prescaled) {
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = pack2<T>(lo_f<T>(raw[j]) * cq, hi_f<T>(raw[j]) * cq);
      }
      qf[qb][ds] = as_vec8<T>(raw);
    }
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[qb][db][i] = 0.f;
    m_run[qb] = 0.f;  // the softmax reference is kept FINITE: the first tile moves it to that tile's row maximum whatever its sign
    l_run[qb] = 0.f;
    if (p.state_in) {
      const int64_t srow = (int64_t)b * p.tq + qrow[qb];
      const float* so = p.st_o + srow * ((int64_t)p.n_heads * HD) + (int64_t)head * HD;
      const float* sm = p.st_ml + (srow * p.n_heads + head) * 4;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int d = db * 32 + 8 * rq + 4 * g;
          if (d < HD) {
            const float4v v = *(const float4v*)(so + d);
            o[qb][db][rq * 4 + 0] = v[0]; o[qb][db][rq * 4 + 1] = v[1]; o[qb][db][rq * 4 + 2] = v[2]; o[qb][db][rq * 4 + 3] = v[3];
          }
        }
      m_run[qb] = sm[0];
      l_run[qb] = sm[1 + g];
    }
  }
  const int krow_pi = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);

  // ---- the tile stream: (segment, first key) pairs in order; `stage` holds the global loads of the tile that is computed NEXT
  u32x4 stage_k[NLD], stage_v[NLD];
  auto first_tile = [&](int& sg, int64_t& k0) {  // first non-empty segment at or after sg
    while (sg < p.n_seg && p.seg_len[sg] <= 0) ++sg;
    k0 = 0;
    return sg < p.n_seg;
  };
  auto next_tile = [&](int& sg, int64_t& k0) {
    k0 += 64;
    if (k0 < p.seg_len[sg]) return true;
    ++sg;
    return first_tile(sg, k0);
  };
  auto load_tile = [&](int sg, int64_t k0) {
    const int64_t n_keys = p.seg_len[sg];
    const int valid = (int)(n_keys - k0 < 64 ? n_keys - k0 : 64);
    const uint16_t* Kg = (const uint16_t*)p.k_seg[sg] + (int64_t)b * p.k_batch_stride[sg] + (int64_t)kv_head * HD;
    const uint16_t* Vg = (const uint16_t*)p.vt_seg[sg] + (int64_t)b * p.vt_batch_stride[sg] + (int64_t)kv_head * HD * p.ldvt[sg];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + i * 256;
      if (c < NCH) {
        const int row = c / (HD / 8), ch = c % (HD / 8);  // K tile: 64 rows x HD/8 chunks of 16 bytes
        const int krow = row < valid ? row : valid - 1;
        stage_k[i] = *(const u32x4*)(Kg + (k0 + krow) * p.ldk + ch * 8);
        const int vrow = c >> 3, vch = c & 7;             // V^T tile: HD rows x 8 chunks (rows zero padded to ldvt, a multiple of 64)
        stage_v[i] = *(const u32x4*)(Vg + (int64_t)vrow * p.ldvt[sg] + k0 + vch * 8);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + i * 256;
      if (c < NCH) {
        *(u32x4*)(kt + (c / (HD / 8)) * KLD + (c % (HD / 8)) * 8) = stage_k[i];
        *(u32x4*)(vt + (c >> 3) * VLD + (c & 7) * 8) = stage_v[i];
      }
    }
  };

  bool first = !p.state_in;  // (a resumed state already carries a reference that only moves up)
  int sg = 0;
  int64_t k0 = 0;
  bool have = first_tile(sg, k0);
  if (have) load_tile(sg, k0);
  while (have) {
    const int valid = (int)(p.seg_len[sg] - k0 < 64 ? p.seg_len[sg] - k0 : 64);
    __syncthreads();  // the previous tile is consumed
    store_tile();
    __syncthreads();
    have = next_tile(sg, k0);
    if (have) load_tile(sg, k0);  // in flight while this tile is computed
    // ---- S^T = K Q^T - m (exp2 units): a K fragment feeds both query blocks; the reference enters as the C operand of the first k-step
    float16v s[QB][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[qb][kb][i] = -m_run[qb];
#pragma unroll
      for (int ds = 0; ds < KS; ++ds) {
        const typename T::vec8 a = as_vec8<T>(*(const u32x4*)(kt + (kb * 32 + krow_pi) * KLD + ds * 16 + g * 8));
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s[qb][kb] = T::mfma32(a, qf[qb][ds], s[qb][kb]);
      }
    }
    typename T::vec8 pf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      // register r of block kb is key kb*32 + 16*(r>>3) + 8*g + (r&7)
      if (valid < 64) {  // the last tile of a segment
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kb * 32 + 16 * (r >> 3) + 8 * g + (r & 7) >= valid) s[qb][kb][r] = -INFINITY;
      }
      float tmax = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[qb][kb][r]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));  // finite: every tile holds a valid key
      // scores are relative to the running reference: it moves (up) only when some row has a positive one
      const float delta = first ? tmax : fmaxf(tmax, 0.f);
      if (first || __builtin_amdgcn_ballot_w64(delta > 0.f) != 0) {
        m_run[qb] += delta;
        if (!first) {  // (nothing is accumulated yet on the first tile, and 2^-delta may not be representable there)
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          l_run[qb] *= alpha;
#pragma unroll
          for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[qb][db][i] *= alpha;
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[qb][kb][r] -= delta;
      }
      float psum = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        u32x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p0 = __builtin_amdgcn_exp2f(s[qb][ks >> 1][(ks & 1) * 8 + 2 * j]);
          const float p1 = __builtin_amdgcn_exp2f(s[qb][ks >> 1][(ks & 1) * 8 + 2 * j + 1]);
          pk[j] = pack2<T>(p0, p1);
          psum = T::sum2(pk[j], psum);  // the sum of what P V multiplies (the rounded probabilities), two per instruction
        }
        pf[qb][ks] = as_vec8<T>(pk);
      }
      l_run[qb] += psum;
    }
    first = false;
    // ---- O^T += V^T P^T: a V^T fragment feeds both query blocks
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const typename T::vec8 a = as_vec8<T>(*(const u32x4*)(vt + (db * 32 + lq) * VLD + ks * 16 + g * 8));
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) o[qb][db] = T::mfma32(a, pf[qb][ks], o[qb][db]);
      }
  }
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    if (!q_ok[qb]) continue;
    const int64_t srow = (int64_t)b * p.tq + qrow[qb];
    if (p.state_out) {
      float* so = p.st_o + srow * ((int64_t)p.n_heads * HD) + (int64_t)head * HD;
      float* sm = p.st_ml + (srow * p.n_heads + head) * 4;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int d = db * 32 + 8 * rq + 4 * g;
          if (d < HD) {
            float4v v = {o[qb][db][rq * 4 + 0], o[qb][db][rq * 4 + 1], o[qb][db][rq * 4 + 2], o[qb][db][rq * 4 + 3]};
            *(float4v*)(so + d) = v;
          }
        }
      if (g == 0) sm[0] = m_run[qb];
      sm[1 + g] = l_run[qb];
      continue;
    }
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    uint16_t* Og = (uint16_t*)p.o + (int64_t)b * p.o_batch_stride + qrow[qb] * p.ldo + (int64_t)head * HD;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int d = db * 32 + 8 * rq + 4 * g;
        if (d < HD) {
          u32x2 w;
          w[0] = pack2<T>(o[qb][db][rq * 4 + 0] * inv, o[qb][db][rq * 4 + 1] * inv);
          w[1] = pack2<T>(o[qb][db][rq * 4 + 2] * inv, o[qb][db][rq * 4 + 3] * inv);
          *(u32x2*)(Og + d) = w;
        }
      }
  }
}

template <class T, int HD>
int launch_hd(const f3r_attn_args& a, hipStream_t s) {
  constexpr int QB = HD <= F3R_GENERIC_QB2_MAX_HD ? 2 : 1;
  constexpr int GQ = 4 * 32 * QB;
  const int64_t qblocks = (a.tq + GQ - 1) / GQ;
  F3R_REQUIRE(qblocks < (1ll << 31) && a.n_heads < 65536 && a.batch < 65536, "f3r_attn_fwd: grid too large");
  hipLaunchKernelGGL((attn_generic_kernel<T, HD, QB>), dim3((unsigned)qblocks, (unsigned)a.n_heads, (unsigned)a.batch), dim3(256), 0, s, a);
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
