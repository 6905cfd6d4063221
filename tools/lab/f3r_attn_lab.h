// Experimental attention bodies kept for the record of the round-1 variant study.  All of them pass
// the parity tests and all of them measured SLOWER than the product body (attn_kernel in f3r_attn.hip, variant 24); they are
// reachable only through f3r_attn_set_variant() / F3R_ATTN_VARIANT and are compiled as part of f3r_attn.hip.
//   v2  running max through the MFMA C operand, packed row sums                      807-870 TF/s
//   v3  software-pipelined QK^T(t+1) || softmax(t), 3-slot LDS ring                   748-790 TF/s
//   v4  all LDS fragments prefetched one phase ahead                                   800-842 TF/s
//   pp  barrier-enforced ping-pong of a matrix phase and a softmax phase (2 x 4 waves)     791 TF/s
#pragma once

// ------------------------------------------------------------------------------------------------------------
// v2 body: fewer VALU instructions per MFMA (head_dim 64 gives only 16 MFMAs per 32 x 64 score block, half of what
// head_dim 128 kernels get, so the softmax VALU stream -- not the matrix pipe -- is what bounds this kernel).
//   * Q is pre-multiplied by scale*log2(e) (by the QKV GEMM epilogue, or here in the prologue), and the running
//     maximum enters through the MFMA C operand: the first MFMA of every score block accumulates onto 16 registers
//     holding -m (in exp2 units), so the block comes out as s' = (q.k)*c - m and P = exp2(s') needs no per-element
//     subtract/fma.  The 16 registers change only when the running max moves (rare after the first tiles).
//   * "max moved" is detected on s' (any s' > 0); only then are s', O, l re-based (wave-uniform rare branch).
//   * row sums accumulate as packed pairs (v_pk_add_f32).
template <class T, int NW, int QPW, int MINW>
__global__ __launch_bounds__(NW * 64, MINW) void attn_kernel_v2(const f3r_attn_args p) {
  constexpr int NT = NW * 64;
  constexpr int QB = NW * QPW * 32;
  constexpr int CPT = 512 / NT;
  typedef float float2v __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * 2 * AT_TILE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int lq = lane & 31;
  const int g = lane >> 5;
  const int head = blockIdx.y;
  const int b = blockIdx.z;
  const int64_t q0 = (int64_t)blockIdx.x * QB + wid * (QPW * 32);
  const float c = p.scale * 1.44269504088896340736f;

  const uint16_t* Qg = (const uint16_t*)p.q + (int64_t)b * p.q_batch_stride;
  int64_t qrow[QPW];
  bool q_ok[QPW];
  typename T::vec8 qf[QPW][4];
#pragma unroll
  for (int qb = 0; qb < QPW; ++qb) {
    qrow[qb] = q0 + qb * 32 + lq;
    q_ok[qb] = qrow[qb] < p.tq;
    if (!q_ok[qb]) qrow[qb] = p.tq - 1;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      u32x4 raw = *(const u32x4*)(Qg + qrow[qb] * p.ldq + head * 64 + ds * 16 + g * 8);
      if (!p.q_prescaled) {
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = pack2<T>(lo_f<T>(raw[j]) * c, hi_f<T>(raw[j]) * c);
      }
      qf[qb][ds] = as_vec8<T>(raw);
    }
  }

  const int sch = tid & 7;
  const int srow0 = tid >> 3;
  int seg_ld = -1;
  int64_t key_ld = 0, seg_keys = 0, seg_ldvt = 0;
  const uint16_t* Kg = nullptr;
  const uint16_t* Vg = nullptr;
  auto next_segment = [&]() {
    key_ld = 0;
    seg_keys = 0;
    for (++seg_ld; seg_ld < p.n_seg; ++seg_ld)
      if (p.seg_len[seg_ld] > 0) {
        seg_keys = p.seg_len[seg_ld];
        seg_ldvt = p.ldvt[seg_ld];
        Kg = (const uint16_t*)p.k_seg[seg_ld] + (int64_t)b * p.k_batch_stride[seg_ld] + head * 64 + sch * 8;
        Vg = (const uint16_t*)p.vt_seg[seg_ld] + (int64_t)b * p.vt_batch_stride[seg_ld] + (int64_t)head * 64 * seg_ldvt + sch * 8;
        break;
      }
  };
  next_segment();
  u32x4 rk[CPT], rv[CPT];
  int valid_ld = 0;
  auto load_next = [&]() -> bool {
    if (seg_keys == 0) return false;
    const int64_t rem = seg_keys - key_ld;
    valid_ld = rem < AT_KB ? (int)rem : AT_KB;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int srow = srow0 + i * (NT / 8);
      u32x4 z = {0u, 0u, 0u, 0u};
      rk[i] = z;
      if (srow < valid_ld) rk[i] = *(const u32x4*)(Kg + (key_ld + srow) * p.ldk);
      rv[i] = *(const u32x4*)(Vg + (int64_t)srow * seg_ldvt + key_ld);
    }
    key_ld += AT_KB;
    if (key_ld >= seg_keys) next_segment();
    return true;
  };
  auto store_tile = [&](int buf) {
    uint16_t* kt = lds + buf * 2 * AT_TILE;
    uint16_t* vt = kt + AT_TILE;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int srow = srow0 + i * (NT / 8);
      *(u32x4*)(kt + aswz(srow, sch)) = rk[i];
      *(u32x4*)(vt + aswz(srow, sch)) = rv[i];
    }
  };

  float16v o[QPW][2];
  float16v negm[QPW];  // 16 copies of -(running max) in exp2 units: the C operand of the first MFMA of a score block
  float l_run[QPW];
#pragma unroll
  for (int qb = 0; qb < QPW; ++qb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[qb][0][i] = 0.f; o[qb][1][i] = 0.f; negm[qb][i] = 0.f; }
    l_run[qb] = 0.f;
  }
  const int krow_pi = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);

  load_next();
  int valid_cur = valid_ld;
  store_tile(0);
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): Q and tile 0 have landed (see attn_kernel)
  int cur = 0;
  bool have = true;
  bool first = true;
  while (have) {
    const int valid = valid_cur;
    const bool more = load_next();
    const uint16_t* kt = lds + cur * 2 * AT_TILE;
    const uint16_t* vt = kt + AT_TILE;

    float16v s[QPW][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        const typename T::vec8 a = as_vec8<T>(*(const u32x4*)(kt + aswz(kb * 32 + krow_pi, ds * 2 + g)));
#pragma unroll
        for (int qb = 0; qb < QPW; ++qb) s[qb][kb] = T::mfma32(a, qf[qb][ds], ds == 0 ? negm[qb] : s[qb][kb]);
      }
    if (valid < AT_KB) {
#pragma unroll
      for (int qb = 0; qb < QPW; ++qb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + 16 * (r >> 3) + 8 * g + (r & 7);
            if (key >= valid) s[qb][kb][r] = -1e30f;
          }
    }
    typename T::vec8 pf[QPW][4];
#pragma unroll
    for (int qb = 0; qb < QPW; ++qb) {
      // max of s' (relative to the running max): 4 short chains instead of one long one
      float m0 = fmaxf(fmaxf(s[qb][0][0], s[qb][0][1]), s[qb][0][2]);
      float m1 = fmaxf(fmaxf(s[qb][0][8], s[qb][0][9]), s[qb][0][10]);
      float m2 = fmaxf(fmaxf(s[qb][1][0], s[qb][1][1]), s[qb][1][2]);
      float m3 = fmaxf(fmaxf(s[qb][1][8], s[qb][1][9]), s[qb][1][10]);
#pragma unroll
      for (int r = 3; r < 7; r += 2) {
        m0 = fmaxf(fmaxf(m0, s[qb][0][r]), s[qb][0][r + 1]);
        m1 = fmaxf(fmaxf(m1, s[qb][0][8 + r]), s[qb][0][8 + r + 1]);
        m2 = fmaxf(fmaxf(m2, s[qb][1][r]), s[qb][1][r + 1]);
        m3 = fmaxf(fmaxf(m3, s[qb][1][8 + r]), s[qb][1][8 + r + 1]);
      }
      m0 = fmaxf(fmaxf(m0, s[qb][0][7]), m1);
      m2 = fmaxf(fmaxf(m2, s[qb][1][7]), m3);
      float mx = fmaxf(fmaxf(m0, s[qb][0][15]), fmaxf(m2, s[qb][1][15]));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (first || __any(mx > 0.f)) {  // wave-uniform, rare after the first tiles: re-base everything on the new max
        const float delta = first ? mx : fmaxf(mx, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s[qb][0][i] -= delta;
          s[qb][1][i] -= delta;
          o[qb][0][i] *= alpha;
          o[qb][1][i] *= alpha;
        }
        l_run[qb] *= alpha;
        const float nm = negm[qb][0] - delta;
#pragma unroll
        for (int i = 0; i < 16; ++i) negm[qb][i] = nm;
      }
      float2v ps0 = {0.f, 0.f}, ps1 = {0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        u32x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float2v e;
          e[0] = __builtin_amdgcn_exp2f(s[qb][ks >> 1][(ks & 1) * 8 + 2 * j]);
          e[1] = __builtin_amdgcn_exp2f(s[qb][ks >> 1][(ks & 1) * 8 + 2 * j + 1]);
          if (j & 1) ps1 += e; else ps0 += e;
          pk[j] = pack2<T>(e[0], e[1]);
        }
        pf[qb][ks] = as_vec8<T>(pk);
      }
      ps0 += ps1;
      l_run[qb] += ps0[0] + ps0[1];
    }
    first = false;

#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const typename T::vec8 a = as_vec8<T>(*(const u32x4*)(vt + aswz(db * 32 + lq, ks * 2 + g)));
#pragma unroll
        for (int qb = 0; qb < QPW; ++qb) o[qb][db] = T::mfma32(a, pf[qb][ks], o[qb][db]);
      }

    if (more) store_tile(cur ^ 1);
    valid_cur = valid_ld;
    __syncthreads();
    cur ^= 1;
    have = more;
  }

#pragma unroll
  for (int qb = 0; qb < QPW; ++qb) {
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (q_ok[qb]) {
      uint16_t* Og = (uint16_t*)p.o + (int64_t)b * p.o_batch_stride + qrow[qb] * p.ldo + head * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          u32x2 w;
          w[0] = pack2<T>(o[qb][db][rq * 4 + 0] * inv, o[qb][db][rq * 4 + 1] * inv);
          w[1] = pack2<T>(o[qb][db][rq * 4 + 2] * inv, o[qb][db][rq * 4 + 3] * inv);
          *(u32x2*)(Og + db * 32 + 8 * rq + 4 * g) = w;
        }
    }
  }
}

template <class T, int NW, int QPW, int MINW>
int attn_launch_v2(const f3r_attn_args& a, hipStream_t s) {
  constexpr int QB = NW * QPW * 32;
  const int64_t qblocks = (a.tq + QB - 1) / QB;
  F3R_REQUIRE(qblocks < (1ll << 31) && a.n_heads < 65536 && a.batch < 65536, "f3r_attn_fwd: grid too large");
  dim3 grid((unsigned)qblocks, (unsigned)a.n_heads, (unsigned)a.batch);
  hipLaunchKernelGGL((attn_kernel_v2<T, NW, QPW, MINW>), grid, dim3(NW * 64), 0, s, a);
  return f3r_check_launch("f3r_attn_fwd");
}

// ------------------------------------------------------------------------------------------------------------
// v3 body: software-pipelined.  In v1/v2 a wave runs strictly phased (QK^T MFMAs -> softmax VALU -> PV MFMAs) and
// the matrix pipe idles during its softmax (measured: no-softmax ablation 1.30 PF/s vs 0.92 PF/s with softmax).
// An MFMA only costs its wave one issue slot; the other ~28 of its 32 pipe cycles are free for independent VALU of
// the SAME wave.  So iteration t issues the QK^T MFMAs of tile t+1 (into a second S accumulator set) interleaved
// with the softmax VALU of tile t, then the P V MFMAs of tile t interleaved with the remaining exp/convert work.
// LDS is a 3-slot ring: iteration t reads K of tile t+1 and V^T of tile t while tile t+2 lands in the third slot.
// Q arrives pre-multiplied by scale*log2(e) (or is scaled in the prologue), scores live in exp2 units.
template <class T, int NW, int MINW, int SGB>
__global__ __launch_bounds__(NW * 64, MINW) void attn_kernel_v3(const f3r_attn_args p) {
  constexpr int NT = NW * 64;
  constexpr int QB = NW * 32;
  constexpr int CPT = 512 / NT;
  typedef float float2v __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) uint16_t lds[3 * 2 * AT_TILE];  // 3 slots x [K | Vt] x 8 KB = 48 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int lq = lane & 31;
  const int g = lane >> 5;
  const int head = blockIdx.y;
  const int b = blockIdx.z;
  const int64_t q0 = (int64_t)blockIdx.x * QB + wid * 32;
  const float c = p.scale * 1.44269504088896340736f;

  const uint16_t* Qg = (const uint16_t*)p.q + (int64_t)b * p.q_batch_stride;
  int64_t qrow = q0 + lq;
  const bool q_ok = qrow < p.tq;
  if (!q_ok) qrow = p.tq - 1;
  typename T::vec8 qf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    u32x4 raw = *(const u32x4*)(Qg + qrow * p.ldq + head * 64 + ds * 16 + g * 8);
    if (!p.q_prescaled) {
#pragma unroll
      for (int j = 0; j < 4; ++j) raw[j] = pack2<T>(lo_f<T>(raw[j]) * c, hi_f<T>(raw[j]) * c);
    }
    qf[ds] = as_vec8<T>(raw);
  }

  const int sch = tid & 7;
  const int srow0 = tid >> 3;
  int seg_ld = -1;
  int64_t key_ld = 0, seg_keys = 0, seg_ldvt = 0;
  const uint16_t* Kg = nullptr;
  const uint16_t* Vg = nullptr;
  auto next_segment = [&]() {
    key_ld = 0;
    seg_keys = 0;
    for (++seg_ld; seg_ld < p.n_seg; ++seg_ld)
      if (p.seg_len[seg_ld] > 0) {
        seg_keys = p.seg_len[seg_ld];
        seg_ldvt = p.ldvt[seg_ld];
        Kg = (const uint16_t*)p.k_seg[seg_ld] + (int64_t)b * p.k_batch_stride[seg_ld] + head * 64 + sch * 8;
        Vg = (const uint16_t*)p.vt_seg[seg_ld] + (int64_t)b * p.vt_batch_stride[seg_ld] + (int64_t)head * 64 * seg_ldvt + sch * 8;
        break;
      }
  };
  next_segment();
  u32x4 rk[CPT], rv[CPT];
  int valid_ld = 0;
  auto load_next = [&]() -> bool {
    if (seg_keys == 0) return false;
    const int64_t rem = seg_keys - key_ld;
    valid_ld = rem < AT_KB ? (int)rem : AT_KB;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int srow = srow0 + i * (NT / 8);
      u32x4 z = {0u, 0u, 0u, 0u};
      rk[i] = z;
      if (srow < valid_ld) rk[i] = *(const u32x4*)(Kg + (key_ld + srow) * p.ldk);
      rv[i] = *(const u32x4*)(Vg + (int64_t)srow * seg_ldvt + key_ld);
    }
    key_ld += AT_KB;
    if (key_ld >= seg_keys) next_segment();
    return true;
  };
  auto store_tile = [&](int slot) {
    uint16_t* kt = lds + slot * 2 * AT_TILE;
    uint16_t* vt = kt + AT_TILE;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int srow = srow0 + i * (NT / 8);
      *(u32x4*)(kt + aswz(srow, sch)) = rk[i];
      *(u32x4*)(vt + aswz(srow, sch)) = rv[i];
    }
  };
  const int krow_pi = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);
  // per-lane LDS element offsets of the fragments (slot base added per use)
  int koff[2][4], voff[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      koff[i][j] = aswz(i * 32 + krow_pi, j * 2 + g);
      voff[i][j] = AT_TILE + aswz(i * 32 + lq, j * 2 + g);
    }

  auto qk = [&](float16v (&sn)[2], int slot) {  // S^T = K Q^T of the tile in `slot`
    const uint16_t* kt = lds + slot * 2 * AT_TILE;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sn[kb][i] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds)
        sn[kb] = T::mfma32(as_vec8<T>(*(const u32x4*)(kt + koff[kb][ds])), qf[ds], sn[kb]);
    }
  };

  float16v o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;

  // ---- pipeline prologue: tiles 0 and 1 staged, S(0) computed
  load_next();
  int valid0 = valid_ld;
  store_tile(0);
  bool have1 = load_next();
  int valid1 = valid_ld;
  if (have1) store_tile(1);
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): Q, tile 0, tile 1 have landed (see attn_kernel)
  float16v sA[2], sB[2];
  qk(sA, 0);

  // one pipeline stage: softmax + PV of the tile in slot `st` (scores in `sc`, `vc` valid keys), QK^T of the next tile
  // (slot `sn1`, if `have_next`) into `sx`, prefetch of tile t+2 into slot `sn2`.
  auto stage = [&](auto has_next_tag, float16v (&sc)[2], float16v (&sx)[2], int st, int sn1, int sn2, int vc) -> bool {
    constexpr bool have_next = decltype(has_next_tag)::value;  // compile-time: keeps the steady-state stage ONE basic block
    const bool more = have_next ? load_next() : false;  // tile t+2 -> registers
    if (vc < AT_KB) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + 16 * (r >> 3) + 8 * g + (r & 7);
          if (key >= vc) sc[kb][r] = -1e30f;
        }
    }
    // QK^T of tile t+1 first in program order; the scheduler hints below spread these MFMAs over the softmax VALU
    if (have_next) qk(sx, sn1);
    // row max: 4 short chains
    float m0 = fmaxf(fmaxf(sc[0][0], sc[0][1]), sc[0][2]);
    float m1 = fmaxf(fmaxf(sc[0][8], sc[0][9]), sc[0][10]);
    float m2 = fmaxf(fmaxf(sc[1][0], sc[1][1]), sc[1][2]);
    float m3 = fmaxf(fmaxf(sc[1][8], sc[1][9]), sc[1][10]);
#pragma unroll
    for (int r = 3; r < 7; r += 2) {
      m0 = fmaxf(fmaxf(m0, sc[0][r]), sc[0][r + 1]);
      m1 = fmaxf(fmaxf(m1, sc[0][8 + r]), sc[0][8 + r + 1]);
      m2 = fmaxf(fmaxf(m2, sc[1][r]), sc[1][r + 1]);
      m3 = fmaxf(fmaxf(m3, sc[1][8 + r]), sc[1][8 + r + 1]);
    }
    m0 = fmaxf(fmaxf(m0, sc[0][7]), m1);
    m2 = fmaxf(fmaxf(m2, sc[1][7]), m3);
    float mx = fmaxf(fmaxf(m0, sc[0][15]), fmaxf(m2, sc[1][15]));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
    float2v ps0 = {0.f, 0.f}, ps1 = {0.f, 0.f};
    const uint16_t* vt = lds + st * 2 * AT_TILE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 pk;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float2v e;
        e[0] = __builtin_amdgcn_exp2f(sc[ks >> 1][(ks & 1) * 8 + 2 * j] - m_new);
        e[1] = __builtin_amdgcn_exp2f(sc[ks >> 1][(ks & 1) * 8 + 2 * j + 1] - m_new);
        if (j & 1) ps1 += e; else ps0 += e;
        pk[j] = pack2<T>(e[0], e[1]);
      }
      const typename T::vec8 pf = as_vec8<T>(pk);
#pragma unroll
      for (int db = 0; db < 2; ++db)
        o[db] = T::mfma32(as_vec8<T>(*(const u32x4*)(vt + voff[db][ks])), pf, o[db]);
    }
    ps0 += ps1;
    l_run = l_run * alpha + (ps0[0] + ps0[1]);
    if (SGB) {
      // desired issue order: every MFMA followed by a few LDS reads and a slice of VALU, 16 times
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
        __builtin_amdgcn_sched_group_barrier(0x002, SGB, 0); // SGB VALU
      }
    }
    if (more) store_tile(sn2);
    __syncthreads();
    return more;
  };

  // ---- steady state, unrolled by two so that the S accumulator sets swap roles without copies.  Slots rotate 0,1,2.
  bool have_next = have1;  // tile t+1 exists (staged)
  int vc = valid0, vn = valid1;
  int st = 0;
  for (;;) {
    const int s1 = st == 2 ? 0 : st + 1, s2 = s1 == 2 ? 0 : s1 + 1;
    if (!have_next) { stage(std::false_type{}, sA, sB, st, s1, s2, vc); break; }
    bool more = stage(std::true_type{}, sA, sB, st, s1, s2, vc);
    vc = vn; vn = valid_ld; have_next = more; st = s1;
    const int t1 = st == 2 ? 0 : st + 1, t2 = t1 == 2 ? 0 : t1 + 1;
    if (!have_next) { stage(std::false_type{}, sB, sA, st, t1, t2, vc); break; }
    more = stage(std::true_type{}, sB, sA, st, t1, t2, vc);
    vc = vn; vn = valid_ld; have_next = more; st = t1;
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (q_ok) {
    uint16_t* Og = (uint16_t*)p.o + (int64_t)b * p.o_batch_stride + qrow * p.ldo + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        u32x2 w;
        w[0] = pack2<T>(o[db][rq * 4 + 0] * inv, o[db][rq * 4 + 1] * inv);
        w[1] = pack2<T>(o[db][rq * 4 + 2] * inv, o[db][rq * 4 + 3] * inv);
        *(u32x2*)(Og + db * 32 + 8 * rq + 4 * g) = w;
      }
  }
}

template <class T, int NW, int MINW, int SGB>
int attn_launch_v3(const f3r_attn_args& a, hipStream_t s) {
  constexpr int QB = NW * 32;
  const int64_t qblocks = (a.tq + QB - 1) / QB;
  F3R_REQUIRE(qblocks < (1ll << 31) && a.n_heads < 65536 && a.batch < 65536, "f3r_attn_fwd: grid too large");
  dim3 grid((unsigned)qblocks, (unsigned)a.n_heads, (unsigned)a.batch);
  hipLaunchKernelGGL((attn_kernel_v3<T, NW, MINW, SGB>), grid, dim3(NW * 64), 0, s, a);
  return f3r_check_launch("f3r_attn_fwd");
}

// ------------------------------------------------------------------------------------------------------------
// v4 body: phased like v1 (QK^T -> softmax -> PV per tile, one S set) but with every LDS fragment read hoisted a
// phase ahead.  Measured on v1: without softmax the kernel still only reaches 52 % of the MFMA roof because each
// "ds_read_b128 -> s_waitcnt -> 2 MFMA" step exposes the LDS latency.  Here the 8 K fragments of tile t+1 are read
// while P V of tile t runs, and the 8 V^T fragments of tile t are read before its softmax, so both MFMA phases issue
// back to back from registers.  That needs tile t+1 resident one iteration early: 3-slot LDS ring as in v3.
// OPT bit 3: ABLATION (timing only) no softmax.
template <class T, int NW, int MINW, int OPT>
__global__ __launch_bounds__(NW * 64, MINW) void attn_kernel_v4(const f3r_attn_args p) {
  constexpr int NT = NW * 64;
  constexpr int QB = NW * 32;
  constexpr int CPT = 512 / NT;
  typedef float float2v __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) uint16_t lds[3 * 2 * AT_TILE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int lq = lane & 31;
  const int g = lane >> 5;
  const int head = blockIdx.y;
  const int b = blockIdx.z;
  const int64_t q0 = (int64_t)blockIdx.x * QB + wid * 32;
  const float c = p.scale * 1.44269504088896340736f;

  const uint16_t* Qg = (const uint16_t*)p.q + (int64_t)b * p.q_batch_stride;
  int64_t qrow = q0 + lq;
  const bool q_ok = qrow < p.tq;
  if (!q_ok) qrow = p.tq - 1;
  typename T::vec8 qf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    u32x4 raw = *(const u32x4*)(Qg + qrow * p.ldq + head * 64 + ds * 16 + g * 8);
    if (!p.q_prescaled) {
#pragma unroll
      for (int j = 0; j < 4; ++j) raw[j] = pack2<T>(lo_f<T>(raw[j]) * c, hi_f<T>(raw[j]) * c);
    }
    qf[ds] = as_vec8<T>(raw);
  }

  const int sch = tid & 7;
  const int srow0 = tid >> 3;
  int seg_ld = -1;
  int64_t key_ld = 0, seg_keys = 0, seg_ldvt = 0;
  const uint16_t* Kg = nullptr;
  const uint16_t* Vg = nullptr;
  auto next_segment = [&]() {
    key_ld = 0;
    seg_keys = 0;
    for (++seg_ld; seg_ld < p.n_seg; ++seg_ld)
      if (p.seg_len[seg_ld] > 0) {
        seg_keys = p.seg_len[seg_ld];
        seg_ldvt = p.ldvt[seg_ld];
        Kg = (const uint16_t*)p.k_seg[seg_ld] + (int64_t)b * p.k_batch_stride[seg_ld] + head * 64 + sch * 8;
        Vg = (const uint16_t*)p.vt_seg[seg_ld] + (int64_t)b * p.vt_batch_stride[seg_ld] + (int64_t)head * 64 * seg_ldvt + sch * 8;
        break;
      }
  };
  next_segment();
  u32x4 rk[CPT], rv[CPT];
  int valid_ld = 0;
  auto load_next = [&]() -> bool {
    if (seg_keys == 0) return false;
    const int64_t rem = seg_keys - key_ld;
    valid_ld = rem < AT_KB ? (int)rem : AT_KB;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int srow = srow0 + i * (NT / 8);
      u32x4 z = {0u, 0u, 0u, 0u};
      rk[i] = z;
      if (srow < valid_ld) rk[i] = *(const u32x4*)(Kg + (key_ld + srow) * p.ldk);
      rv[i] = *(const u32x4*)(Vg + (int64_t)srow * seg_ldvt + key_ld);
    }
    key_ld += AT_KB;
    if (key_ld >= seg_keys) next_segment();
    return true;
  };
  auto store_tile = [&](int slot) {
    uint16_t* kt = lds + slot * 2 * AT_TILE;
    uint16_t* vt = kt + AT_TILE;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int srow = srow0 + i * (NT / 8);
      *(u32x4*)(kt + aswz(srow, sch)) = rk[i];
      *(u32x4*)(vt + aswz(srow, sch)) = rv[i];
    }
  };
  const int krow_pi = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);
  int koff[2][4], voff[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      koff[i][j] = aswz(i * 32 + krow_pi, j * 2 + g);
      voff[i][j] = AT_TILE + aswz(i * 32 + lq, j * 2 + g);
    }
  u32x4 kf[2][4], vf[2][4];
  auto read_k = [&](int slot) {
    const uint16_t* t = lds + slot * 2 * AT_TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) kf[i][j] = *(const u32x4*)(t + koff[i][j]);
  };
  auto read_v = [&](int slot) {
    const uint16_t* t = lds + slot * 2 * AT_TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) vf[i][j] = *(const u32x4*)(t + voff[i][j]);
  };

  float16v o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;

  load_next();
  int vc = valid_ld;  // valid keys of tile t
  store_tile(0);
  bool have_next = load_next();
  int vn = valid_ld;
  if (have_next) store_tile(1);
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): Q, tiles 0 and 1 have landed (see attn_kernel)
  read_k(0);
  int st = 0;
  for (;;) {
    const int s1 = st == 2 ? 0 : st + 1, s2 = s1 == 2 ? 0 : s1 + 1;
    const bool more = have_next ? load_next() : false;  // tile t+2 -> registers
    read_v(st);                                         // V^T fragments of tile t: consumed after the softmax
    float16v s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) s[kb] = T::mfma32(as_vec8<T>(kf[kb][ds]), qf[ds], s[kb]);
    }
    if (have_next) read_k(s1);  // K fragments of tile t+1 (resident since the previous barrier): consumed next iteration
    if (vc < AT_KB) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + 16 * (r >> 3) + 8 * g + (r & 7);
          if (key >= vc) s[kb][r] = -1e30f;
        }
    }
    typename T::vec8 pf[4];
    if (OPT & 8) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        u32x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[j] = pack2<T>(s[ks >> 1][(ks & 1) * 8 + 2 * j], s[ks >> 1][(ks & 1) * 8 + 2 * j + 1]);
        pf[ks] = as_vec8<T>(pk);
      }
    } else {
      float m0 = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
      float m1 = fmaxf(fmaxf(s[0][8], s[0][9]), s[0][10]);
      float m2 = fmaxf(fmaxf(s[1][0], s[1][1]), s[1][2]);
      float m3 = fmaxf(fmaxf(s[1][8], s[1][9]), s[1][10]);
#pragma unroll
      for (int r = 3; r < 7; r += 2) {
        m0 = fmaxf(fmaxf(m0, s[0][r]), s[0][r + 1]);
        m1 = fmaxf(fmaxf(m1, s[0][8 + r]), s[0][8 + r + 1]);
        m2 = fmaxf(fmaxf(m2, s[1][r]), s[1][r + 1]);
        m3 = fmaxf(fmaxf(m3, s[1][8 + r]), s[1][8 + r + 1]);
      }
      m0 = fmaxf(fmaxf(m0, s[0][7]), m1);
      m2 = fmaxf(fmaxf(m2, s[1][7]), m3);
      float mx = fmaxf(fmaxf(m0, s[0][15]), fmaxf(m2, s[1][15]));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      if (__any(m_new > m_run)) {  // exact skip: when no running max of the wave moved, alpha == 1
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
        m_run = m_new;
      }
      float2v ps0 = {0.f, 0.f}, ps1 = {0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        u32x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float2v e;
          e[0] = __builtin_amdgcn_exp2f(s[ks >> 1][(ks & 1) * 8 + 2 * j] - m_run);
          e[1] = __builtin_amdgcn_exp2f(s[ks >> 1][(ks & 1) * 8 + 2 * j + 1] - m_run);
          if (j & 1) ps1 += e; else ps0 += e;
          pk[j] = pack2<T>(e[0], e[1]);
        }
        pf[ks] = as_vec8<T>(pk);
      }
      ps0 += ps1;
      l_run += ps0[0] + ps0[1];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int db = 0; db < 2; ++db) o[db] = T::mfma32(as_vec8<T>(vf[db][ks]), pf[ks], o[db]);

    if (more) store_tile(s2);
    __syncthreads();
    if (!have_next) break;
    vc = vn; vn = valid_ld; have_next = more; st = s1;
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (q_ok) {
    uint16_t* Og = (uint16_t*)p.o + (int64_t)b * p.o_batch_stride + qrow * p.ldo + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        u32x2 w;
        w[0] = pack2<T>(o[db][rq * 4 + 0] * inv, o[db][rq * 4 + 1] * inv);
        w[1] = pack2<T>(o[db][rq * 4 + 2] * inv, o[db][rq * 4 + 3] * inv);
        *(u32x2*)(Og + db * 32 + 8 * rq + 4 * g) = w;
      }
  }
}

template <class T, int NW, int MINW, int OPT>
int attn_launch_v4(const f3r_attn_args& a, hipStream_t s) {
  constexpr int QB = NW * 32;
  const int64_t qblocks = (a.tq + QB - 1) / QB;
  F3R_REQUIRE(qblocks < (1ll << 31) && a.n_heads < 65536 && a.batch < 65536, "f3r_attn_fwd: grid too large");
  dim3 grid((unsigned)qblocks, (unsigned)a.n_heads, (unsigned)a.batch);
  hipLaunchKernelGGL((attn_kernel_v4<T, NW, MINW, OPT>), grid, dim3(NW * 64), 0, s, a);
  return f3r_check_launch("f3r_attn_fwd");
}

// ------------------------------------------------------------------------------------------------------------
// Ping-pong body.  Measured on v1: the time of a tile is (MFMA phases) + (softmax phase) -- the two waves that share
// a SIMD do not drift into complementary phases by themselves.  Here the complement is enforced: a workgroup is 8
// waves = two groups of 4 (one wave of each group per SIMD), every wave owns 64 queries, and the loop is cut into
// PHASES separated by s_barrier.  In each phase one group runs a pure matrix body  M(t) = P V of tile t-1, then
// Q K^T of tile t,  while the other runs the pure VALU body  S(t) = online softmax of tile t;  the groups are offset
// by one phase.  So on every SIMD the matrix pipe always has exactly one wave feeding it and the VALU exactly one
// softmax stream.  All 8 waves share the K / V^T staging: in every phase each thread fetches one 16-byte chunk of the
// item that must be resident two phases later and writes the chunk it fetched during the previous phase:
//   phase 2t   : fetch V^T(t),   write K(t+1)        phase 2t+1 : fetch K(t+2),   write V^T(t)
// K and V^T live in two 2-slot rings (32 KB).  Group 0: M(t) in phase 2t, S(t) in 2t+1;  group 1: S(t-1) in 2t, M(t) in 2t+1.
// OPT bit 3: ABLATION (timing only): no softmax.
template <class T, int OPT>
__global__ __launch_bounds__(512, 2) void attn_kernel_pp(const f3r_attn_args p) {
  typedef float float2v __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) uint16_t lds[4 * AT_TILE];  // K ring [2] | V^T ring [2], 8 KB each
  uint16_t* const kring = lds;
  uint16_t* const vring = lds + 2 * AT_TILE;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;
  const int lq = lane & 31;
  const int g = lane >> 5;
  const int head = blockIdx.y;
  const int b = blockIdx.z;
  const int64_t q0 = (int64_t)blockIdx.x * 512 + wid * 64;
  const float c = p.scale * 1.44269504088896340736f;

  const uint16_t* Qg = (const uint16_t*)p.q + (int64_t)b * p.q_batch_stride;
  int64_t qrow[2];
  bool q_ok[2];
  typename T::vec8 qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    qrow[qb] = q0 + qb * 32 + lq;
    q_ok[qb] = qrow[qb] < p.tq;
    if (!q_ok[qb]) qrow[qb] = p.tq - 1;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      u32x4 raw = *(const u32x4*)(Qg + qrow[qb] * p.ldq + head * 64 + ds * 16 + g * 8);
      if (!p.q_prescaled) {
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = pack2<T>(lo_f<T>(raw[j]) * c, hi_f<T>(raw[j]) * c);
      }
      qf[qb][ds] = as_vec8<T>(raw);
    }
  }

  // ---- tile walkers over the K/V segments (K fetches run one tile ahead of V^T fetches; S needs the valid counts)
  int n_tiles = 0;
  for (int sg = 0; sg < p.n_seg; ++sg) n_tiles += (int)((p.seg_len[sg] + AT_KB - 1) / AT_KB);
  const int sch = tid & 7;
  const int srow = tid >> 3;  // 0..63: one chunk of a 64 x 64 tile per thread
  struct Walker {
    int seg;
    int64_t key, len, ld;
    const uint16_t* base;
  };
  auto seek = [&](Walker& w, bool is_k) {  // move to the next non-empty segment
    w.key = 0;
    w.len = 0;
    for (++w.seg; w.seg < p.n_seg; ++w.seg)
      if (p.seg_len[w.seg] > 0) {
        w.len = p.seg_len[w.seg];
        w.ld = p.ldvt[w.seg];
        w.base = is_k ? (const uint16_t*)p.k_seg[w.seg] + (int64_t)b * p.k_batch_stride[w.seg] + head * 64 + sch * 8
                      : (const uint16_t*)p.vt_seg[w.seg] + (int64_t)b * p.vt_batch_stride[w.seg] + ((int64_t)head * 64 + srow) * w.ld + sch * 8;
        break;
      }
  };
  Walker wk{-1, 0, 0, 0, nullptr}, wv{-1, 0, 0, 0, nullptr};
  seek(wk, true);
  seek(wv, false);
  int vs_seg = -1;
  int64_t vs_key = 0, vs_len = 0;  // valid-count walker for the softmax
  auto next_valid = [&]() -> int {
    if (vs_key >= vs_len) {
      vs_key = 0;
      for (++vs_seg; vs_seg < p.n_seg && p.seg_len[vs_seg] <= 0; ++vs_seg) {}
      vs_len = vs_seg < p.n_seg ? p.seg_len[vs_seg] : 0;
    }
    const int64_t rem = vs_len - vs_key;
    vs_key += AT_KB;
    return rem < AT_KB ? (int)rem : AT_KB;
  };
  u32x4 regk, regv;
  auto fetch_k = [&]() {  // one chunk of the next K tile -> regk
    int64_t r = wk.key + srow;
    if (r >= wk.len) r = wk.len - 1;  // rows past the segment end are masked in the softmax: any finite data will do
    regk = *(const u32x4*)(wk.base + r * p.ldk);
    wk.key += AT_KB;
    if (wk.key >= wk.len) seek(wk, true);
  };
  auto fetch_v = [&]() {  // V^T rows are padded to a multiple of 64 with zeros by the host: always in bounds
    regv = *(const u32x4*)(wv.base + wv.key);
    wv.key += AT_KB;
    if (wv.key >= wv.len) seek(wv, false);
  };
  const int soff = aswz(srow, sch);

  const int krow_pi = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);
  int koff[2][4], voff[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      koff[i][j] = aswz(i * 32 + krow_pi, j * 2 + g);
      voff[i][j] = aswz(i * 32 + lq, j * 2 + g);
    }

  float16v o[2][2], s[2][2];
  typename T::vec8 pf[2][4];
  float m_run[2], l_run[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[qb][0][i] = 0.f; o[qb][1][i] = 0.f; }
    m_run[qb] = -1e30f;
    l_run[qb] = 0.f;
  }

  // matrix bodies: P V of a tile (V^T in ring slot `slot`), Q K^T of a tile (K in ring slot `slot`)
  auto pv = [&](int slot) {
    const uint16_t* vt = vring + slot * AT_TILE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const typename T::vec8 a = as_vec8<T>(*(const u32x4*)(vt + voff[db][ks]));
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) o[qb][db] = T::mfma32(a, pf[qb][ks], o[qb][db]);
      }
  };
  auto qk = [&](int slot) {
    const uint16_t* kt = kring + slot * AT_TILE;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[qb][kb][i] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        const typename T::vec8 a = as_vec8<T>(*(const u32x4*)(kt + koff[kb][ds]));
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) s[qb][kb] = T::mfma32(a, qf[qb][ds], s[qb][kb]);
      }
    }
  };
  // VALU body: online softmax of the tile whose scores are in s -> pf, m, l; O is re-based when a max moved
  auto s_body = [&]() {
    const int vc = next_valid();
    if (vc < AT_KB) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + 16 * (r >> 3) + 8 * g + (r & 7);
            if (key >= vc) s[qb][kb][r] = -1e30f;
          }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      if (OPT & 8) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          u32x4 pk;
#pragma unroll
          for (int j = 0; j < 4; ++j) pk[j] = pack2<T>(s[qb][ks >> 1][(ks & 1) * 8 + 2 * j], s[qb][ks >> 1][(ks & 1) * 8 + 2 * j + 1]);
          pf[qb][ks] = as_vec8<T>(pk);
        }
        continue;
      }
      float m0 = fmaxf(fmaxf(s[qb][0][0], s[qb][0][1]), s[qb][0][2]);
      float m1 = fmaxf(fmaxf(s[qb][0][8], s[qb][0][9]), s[qb][0][10]);
      float m2 = fmaxf(fmaxf(s[qb][1][0], s[qb][1][1]), s[qb][1][2]);
      float m3 = fmaxf(fmaxf(s[qb][1][8], s[qb][1][9]), s[qb][1][10]);
#pragma unroll
      for (int r = 3; r < 7; r += 2) {
        m0 = fmaxf(fmaxf(m0, s[qb][0][r]), s[qb][0][r + 1]);
        m1 = fmaxf(fmaxf(m1, s[qb][0][8 + r]), s[qb][0][8 + r + 1]);
        m2 = fmaxf(fmaxf(m2, s[qb][1][r]), s[qb][1][r + 1]);
        m3 = fmaxf(fmaxf(m3, s[qb][1][8 + r]), s[qb][1][8 + r + 1]);
      }
      m0 = fmaxf(fmaxf(m0, s[qb][0][7]), m1);
      m2 = fmaxf(fmaxf(m2, s[qb][1][7]), m3);
      float mx = fmaxf(fmaxf(m0, s[qb][0][15]), fmaxf(m2, s[qb][1][15]));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[qb], mx);
      if (__any(m_new > m_run[qb])) {  // exact skip: alpha == 1 for every lane otherwise
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
        l_run[qb] *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[qb][0][i] *= alpha; o[qb][1][i] *= alpha; }
        m_run[qb] = m_new;
      }
      const float mr = m_run[qb];
      float2v ps0 = {0.f, 0.f}, ps1 = {0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        u32x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float2v e;
          e[0] = __builtin_amdgcn_exp2f(s[qb][ks >> 1][(ks & 1) * 8 + 2 * j] - mr);
          e[1] = __builtin_amdgcn_exp2f(s[qb][ks >> 1][(ks & 1) * 8 + 2 * j + 1] - mr);
          if (j & 1) ps1 += e; else ps0 += e;
          pk[j] = pack2<T>(e[0], e[1]);
        }
        pf[qb][ks] = as_vec8<T>(pk);
      }
      ps0 += ps1;
      l_run[qb] += ps0[0] + ps0[1];
    }
  };

  // ---- prologue: K(0) resident, K(1) in flight
  fetch_k();
  *(u32x4*)(kring + soff) = regk;
  if (n_tiles > 1) fetch_k();
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): Q and the prologue tiles have landed (see attn_kernel)

  // ---- phase loops.  Every wave executes 2 (n_tiles + 1) barriers.  The loops are specialised per group with the
  // first and last tile peeled so that in steady state s (64 VGPRs) and pf (32) are never live at the same boundary.
  const int T_ = n_tiles;
  auto stage_even = [&](int t) {  // phase 2t staging head: fetch V^T(t)
    if (t < T_) fetch_v();
  };
  auto stage_even_tail = [&](int t) {  // phase 2t staging tail: write K(t+1), barrier
    if (t + 1 < T_) *(u32x4*)(kring + ((t + 1) & 1) * AT_TILE + soff) = regk;
    __syncthreads();
  };
  auto stage_odd = [&](int t) {  // phase 2t+1 staging head: fetch K(t+2)
    if (t + 2 < T_) fetch_k();
  };
  auto stage_odd_tail = [&](int t) {  // phase 2t+1 staging tail: write V^T(t), barrier
    if (t < T_) *(u32x4*)(vring + (t & 1) * AT_TILE + soff) = regv;
    __syncthreads();
  };
  if (grp == 0) {  // M(t) in phase 2t, S(t) in phase 2t+1
    stage_even(0); qk(0); stage_even_tail(0);
    stage_odd(0); s_body(); stage_odd_tail(0);
    for (int t = 1; t < T_; ++t) {
      stage_even(t); pv((t - 1) & 1); qk(t & 1); stage_even_tail(t);
      stage_odd(t); s_body(); stage_odd_tail(t);
    }
    stage_even(T_); pv((T_ - 1) & 1); stage_even_tail(T_);
    stage_odd(T_); stage_odd_tail(T_);
  } else {  // S(t-1) in phase 2t, M(t) in phase 2t+1
    stage_even(0); stage_even_tail(0);
    stage_odd(0); qk(0); stage_odd_tail(0);
    for (int t = 1; t < T_; ++t) {
      stage_even(t); s_body(); stage_even_tail(t);
      stage_odd(t); pv((t - 1) & 1); qk(t & 1); stage_odd_tail(t);
    }
    stage_even(T_); s_body(); stage_even_tail(T_);
    stage_odd(T_); pv((T_ - 1) & 1); stage_odd_tail(T_);
  }

#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (q_ok[qb]) {
      uint16_t* Og = (uint16_t*)p.o + (int64_t)b * p.o_batch_stride + qrow[qb] * p.ldo + head * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          u32x2 w;
          w[0] = pack2<T>(o[qb][db][rq * 4 + 0] * inv, o[qb][db][rq * 4 + 1] * inv);
          w[1] = pack2<T>(o[qb][db][rq * 4 + 2] * inv, o[qb][db][rq * 4 + 3] * inv);
          *(u32x2*)(Og + db * 32 + 8 * rq + 4 * g) = w;
        }
    }
  }
}

template <class T, int OPT>
int attn_launch_pp(const f3r_attn_args& a, hipStream_t s) {
  const int64_t qblocks = (a.tq + 511) / 512;
  F3R_REQUIRE(qblocks < (1ll << 31) && a.n_heads < 65536 && a.batch < 65536, "f3r_attn_fwd: grid too large");
  dim3 grid((unsigned)qblocks, (unsigned)a.n_heads, (unsigned)a.batch);
  hipLaunchKernelGGL((attn_kernel_pp<T, OPT>), grid, dim3(512), 0, s, a);
  return f3r_check_launch("f3r_attn_fwd");
}


// ------------------------------------------------------------------------------------------------------------
// sp body: 3-stage software pipeline with a PINNED issue order.
// Measured facts behind it (tools/ubench/valu_rate.hip, instrumented variant 34): one wave issues a VALU instruction every
// ~5 cycles (v_exp_f32 ~9) no matter how many are independent, an MFMA costs its wave one such slot but occupies the matrix
// pipe for 32 cycles, and in the phased bodies (QK^T -> softmax -> PV) the matrix pipe idles during every softmax.  So the
// softmax of tile t (~175 VALU, ~1000 issue cycles per 32 queries) is interleaved, instruction by instruction, with the 16
// MFMAs that do not depend on it: P V of tile t-1 (P already packed) and Q K^T of tile t+1 (second accumulator set).  The
// order is written out by hand as 16 slots { fragment read for slot+2 ; 1 MFMA ; a slice of the softmax } separated by
// __builtin_amdgcn_sched_barrier(0), because the compiler's own schedule clusters the MFMAs in front of the VALU (v3).
// LDS: ring of 4 [K | V^T] tile slots (64 KB): iteration t reads V^T(t-1) and K(t+1) while tile t+2 lands and t+3 is in flight.
#define F3R_SB() __builtin_amdgcn_sched_barrier(0)
template <class T, int NW, int PROF>
__global__ __launch_bounds__(NW * 64, 2) void attn_kernel_sp(const f3r_attn_args p) {
  unsigned long long ta_ = 0, tb_ = 0, tc_ = 0, td_ = 0, nt_ = 0;
  constexpr int NT = NW * 64;
  constexpr int QB = NW * 32;
  constexpr int CPT = 512 / NT;
  // PROF == 2: occupancy experiment -- pad LDS past 80 KB so that only ONE workgroup fits on a CU
  __shared__ __attribute__((aligned(16))) uint16_t lds[(PROF == 2 ? 6 : 4) * 2 * AT_TILE];  // 4 slots x [K | Vt] x 8 KB = 64 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int lq = lane & 31;
  const int g = lane >> 5;
  const int head = blockIdx.y;
  const int b = blockIdx.z;
  const int64_t q0 = (int64_t)blockIdx.x * QB + wid * 32;
  const float c = p.scale * 1.44269504088896340736f;
  if (PROF == 2 && p.tq < 0) lds[5 * 2 * AT_TILE + tid] = 0;  // keep the padding allocated

  const uint16_t* Qg = (const uint16_t*)p.q + (int64_t)b * p.q_batch_stride;
  int64_t qrow = q0 + lq;
  const bool q_ok = qrow < p.tq;
  if (!q_ok) qrow = p.tq - 1;
  typename T::vec8 qf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    u32x4 raw = *(const u32x4*)(Qg + qrow * p.ldq + head * 64 + ds * 16 + g * 8);
    if (!p.q_prescaled) {
#pragma unroll
      for (int j = 0; j < 4; ++j) raw[j] = pack2<T>(lo_f<T>(raw[j]) * c, hi_f<T>(raw[j]) * c);
    }
    qf[ds] = as_vec8<T>(raw);
  }

  // ---- tile walker (loads) and valid-count walker (softmax), as in the other bodies
  int n_tiles = 0;
  for (int sg = 0; sg < p.n_seg; ++sg) n_tiles += (int)((p.seg_len[sg] + AT_KB - 1) / AT_KB);
  const int sch = tid & 7;
  const int srow0 = tid >> 3;
  int seg_ld = -1;
  int64_t key_ld = 0, seg_keys = 0, seg_ldvt = 0;
  const uint16_t* Kg = nullptr;
  const uint16_t* Vg = nullptr;
  auto next_segment = [&]() {
    key_ld = 0;
    seg_keys = 0;
    for (++seg_ld; seg_ld < p.n_seg; ++seg_ld)
      if (p.seg_len[seg_ld] > 0) {
        seg_keys = p.seg_len[seg_ld];
        seg_ldvt = p.ldvt[seg_ld];
        Kg = (const uint16_t*)p.k_seg[seg_ld] + (int64_t)b * p.k_batch_stride[seg_ld] + head * 64 + sch * 8;
        Vg = (const uint16_t*)p.vt_seg[seg_ld] + (int64_t)b * p.vt_batch_stride[seg_ld] + (int64_t)head * 64 * seg_ldvt + sch * 8;
        break;
      }
  };
  next_segment();
  u32x4 rk[CPT], rv[CPT];
  auto load_next = [&]() -> bool {
    if (seg_keys == 0) return false;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int srow = srow0 + i * (NT / 8);
      int64_t r = key_ld + srow;
      if (r >= seg_keys) r = seg_keys - 1;  // rows past the end are masked in the softmax
      rk[i] = *(const u32x4*)(Kg + r * p.ldk);
      rv[i] = *(const u32x4*)(Vg + (int64_t)srow * seg_ldvt + key_ld);
    }
    key_ld += AT_KB;
    if (key_ld >= seg_keys) next_segment();
    return true;
  };
  auto store_tile = [&](int slot) {
    uint16_t* kt = lds + slot * 2 * AT_TILE;
    uint16_t* vt = kt + AT_TILE;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int srow = srow0 + i * (NT / 8);
      *(u32x4*)(kt + aswz(srow, sch)) = rk[i];
      *(u32x4*)(vt + aswz(srow, sch)) = rv[i];
    }
  };
  int vs_seg = -1;
  int64_t vs_key = 0, vs_len = 0;
  auto next_valid = [&]() -> int {
    if (vs_key >= vs_len) {
      vs_key = 0;
      for (++vs_seg; vs_seg < p.n_seg && p.seg_len[vs_seg] <= 0; ++vs_seg) {}
      vs_len = vs_seg < p.n_seg ? p.seg_len[vs_seg] : 0;
    }
    const int64_t rem = vs_len - vs_key;
    vs_key += AT_KB;
    return rem < AT_KB ? (int)rem : AT_KB;
  };

  const int krow_pi = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);
  int koff[2][4], voff[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      koff[i][j] = aswz(i * 32 + krow_pi, j * 2 + g);
      voff[i][j] = AT_TILE + aswz(i * 32 + lq, j * 2 + g);
    }

  float16v o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;
  float16v sA[2], sB[2];
  typename T::vec8 pA[4], pB[4];

  // ---- prologue: tiles 0, 1 resident, tile 2 in registers, S(0) computed
  load_next();
  store_tile(0);
  if (load_next()) store_tile(1);
  bool pend = load_next();  // registers hold tile 2
  __syncthreads();
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see attn_kernel
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) sA[kb][i] = 0.f;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) sA[kb] = T::mfma32(as_vec8<T>(*(const u32x4*)(lds + koff[kb][ds])), qf[ds], sA[kb]);
  }

  // one pipeline stage = iteration t.  sc: scores of tile t; sn: receives scores of tile t+1; pp: P(t-1); pc: receives P(t)
  auto stage = [&](auto has_pv_tag, auto has_qk_tag, float16v (&sc)[2], float16v (&sn)[2], typename T::vec8 (&pp)[4],
                   typename T::vec8 (&pc)[4], int t) {
    constexpr bool HAS_PV = decltype(has_pv_tag)::value;
    constexpr bool HAS_QK = decltype(has_qk_tag)::value;
    const int vc = next_valid();
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (PROF == 1) c0 = __builtin_readcyclecounter();
    if (vc < AT_KB) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + 16 * (r >> 3) + 8 * g + (r & 7);
          if (key >= vc) sc[kb][r] = -1e30f;
        }
    }
    const uint16_t* vt = lds + ((t + 3) & 3) * 2 * AT_TILE;  // tile t-1
    const uint16_t* kt = lds + ((t + 1) & 3) * 2 * AT_TILE;  // tile t+1
    u32x4 fr[8], kf[8];
    if (HAS_PV) {
      fr[0] = *(const u32x4*)(vt + voff[0][0]);
      fr[1] = *(const u32x4*)(vt + voff[1][0]);
    }
    F3R_SB();
    // ---- part A: 8 x { P V MFMA of tile t-1 } interleaved with the row max of tile t
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, mx = 0.f, partner = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (HAS_PV) {
        if (i + 2 < 8) fr[i + 2] = *(const u32x4*)(vt + voff[(i + 2) & 1][(i + 2) >> 1]);
        o[i & 1] = T::mfma32(as_vec8<T>(fr[i]), pp[i >> 1], o[i & 1]);
      }
      if (HAS_QK && i >= 6) kf[i - 6] = *(const u32x4*)(kt + koff[0][i - 6]);
      if (i == 0) {
        m0 = fmaxf(fmaxf(sc[0][0], sc[0][1]), sc[0][2]);
        m1 = fmaxf(fmaxf(sc[0][8], sc[0][9]), sc[0][10]);
        m2 = fmaxf(fmaxf(sc[1][0], sc[1][1]), sc[1][2]);
        m3 = fmaxf(fmaxf(sc[1][8], sc[1][9]), sc[1][10]);
      } else if (i == 1 || i == 2) {
        const int r = 2 * i + 1;  // 3, 5
        m0 = fmaxf(fmaxf(m0, sc[0][r]), sc[0][r + 1]);
        m1 = fmaxf(fmaxf(m1, sc[0][8 + r]), sc[0][8 + r + 1]);
        m2 = fmaxf(fmaxf(m2, sc[1][r]), sc[1][r + 1]);
        m3 = fmaxf(fmaxf(m3, sc[1][8 + r]), sc[1][8 + r + 1]);
      } else if (i == 3) {
        m0 = fmaxf(fmaxf(m0, sc[0][7]), m1);
        m2 = fmaxf(fmaxf(m2, sc[1][7]), m3);
        mx = fmaxf(fmaxf(m0, sc[0][15]), fmaxf(m2, sc[1][15]));
      } else if (i == 4) {
        partner = __shfl_xor(mx, 32, 64);
      } else if (i == 6) {
        mx = fmaxf(mx, partner);
      }
      // anchor the slice here: IR-level sinking/hoisting would otherwise move it out of its slot (sched_barrier only pins
      // the machine scheduler)
      asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(mx), "+v"(partner));
      F3R_SB();
    }
    if (PROF == 1) { asm volatile("" :: "v"(o[0][0]), "v"(o[1][15])); c1 = __builtin_readcyclecounter(); }
    // ---- part B (rare after the first tiles): the running max moved -> re-base O and l
    const float m_new = fmaxf(m_run, mx);
    if (__any(m_new > m_run)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
      m_run = m_new;
    }
    const float mr = m_run;
    if (PROF == 1) c2 = __builtin_readcyclecounter();
    F3R_SB();
    // ---- part C: 8 x { Q K^T MFMA of tile t+1 } interleaved with exp / pack / row-sum of tile t
    float ps0 = 0.f, ps1 = 0.f;
    u32x4 pk;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (HAS_QK) {
        if (i + 2 < 8) kf[i + 2] = *(const u32x4*)(kt + koff[(i + 2) >> 2][(i + 2) & 3]);
        if ((i & 3) == 0) {
#pragma unroll
          for (int z = 0; z < 16; ++z) sn[i >> 2][z] = 0.f;
        }
        sn[i >> 2] = T::mfma32(as_vec8<T>(kf[i]), qf[i & 3], sn[i >> 2]);
      }
      const int ks = i >> 1;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = (i & 1) * 2 + jj;
        const float e0 = __builtin_amdgcn_exp2f(sc[ks >> 1][(ks & 1) * 8 + 2 * j] - mr);
        const float e1 = __builtin_amdgcn_exp2f(sc[ks >> 1][(ks & 1) * 8 + 2 * j + 1] - mr);
        ps0 += e0;
        ps1 += e1;
        pk[j] = pack2<T>(e0, e1);
        asm volatile("" : "+v"(ps0), "+v"(ps1), "+v"(pk[j]));  // anchor (see part A)
      }
      if (i & 1) pc[ks] = as_vec8<T>(pk);
      F3R_SB();
    }
    l_run += ps0 + ps1;
    if (PROF == 1) { if (HAS_QK) asm volatile("" :: "v"(sn[0][0]), "v"(sn[1][15])); c3 = __builtin_readcyclecounter(); }
    // ---- part D: tile t+2 -> LDS, tile t+3 -> registers
    if (pend) store_tile((t + 2) & 3);
    pend = pend ? load_next() : false;
    __syncthreads();
    if (PROF == 1) {
      const unsigned long long c4 = __builtin_readcyclecounter();
      ta_ += c1 - c0; tb_ += c2 - c1; tc_ += c3 - c2; td_ += c4 - c3; ++nt_;
    }
  };

  if (PROF == 3 && NW == 8) {
    if (__builtin_amdgcn_readfirstlane(tid) >= 256) __builtin_amdgcn_s_setprio(1);
  }
  const std::true_type yes{};
  const std::false_type no{};
  // Stage t reads scores from sA when t is even (sB when odd) and leaves P(t) in pA when t is even (pB when odd).  The loop is
  // unrolled by two with fixed roles so that only ONE score set and ONE P set are live across the back-edge.
  if (n_tiles == 1) {
    stage(no, no, sA, sB, pB, pA, 0);
  } else {
    stage(no, yes, sA, sB, pB, pA, 0);
    int t = 1;
    while (t + 2 < n_tiles) {  // neither t nor t+1 is the last tile
      stage(yes, yes, sB, sA, pA, pB, t);
      stage(yes, yes, sA, sB, pB, pA, t + 1);
      t += 2;
    }
    if (t + 1 < n_tiles) {
      stage(yes, yes, sB, sA, pA, pB, t);
      stage(yes, no, sA, sB, pB, pA, t + 1);
    } else {
      stage(yes, no, sB, sA, pA, pB, t);
    }
  }
  if (PROF == 1 && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
    g_attn_prof[0] = ta_; g_attn_prof[1] = tb_; g_attn_prof[2] = tc_; g_attn_prof[3] = td_; g_attn_prof[4] = nt_;
  }
  // ---- drain: P V of the last tile (its P is in pA when (n_tiles - 1) is even, else pB)
  {
    const uint16_t* vt = lds + ((n_tiles - 1) & 3) * 2 * AT_TILE;
    const bool evenlast = ((n_tiles - 1) & 1) == 0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const typename T::vec8 a = as_vec8<T>(*(const u32x4*)(vt + voff[db][ks]));
        if (evenlast) o[db] = T::mfma32(a, pA[ks], o[db]); else o[db] = T::mfma32(a, pB[ks], o[db]);
      }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (q_ok) {
    uint16_t* Og = (uint16_t*)p.o + (int64_t)b * p.o_batch_stride + qrow * p.ldo + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        u32x2 w;
        w[0] = pack2<T>(o[db][rq * 4 + 0] * inv, o[db][rq * 4 + 1] * inv);
        w[1] = pack2<T>(o[db][rq * 4 + 2] * inv, o[db][rq * 4 + 3] * inv);
        *(u32x2*)(Og + db * 32 + 8 * rq + 4 * g) = w;
      }
  }
}

template <class T, int NW, int PROF>
int attn_launch_sp(const f3r_attn_args& a, hipStream_t s) {
  constexpr int QB = NW * 32;
  const int64_t qblocks = (a.tq + QB - 1) / QB;
  F3R_REQUIRE(qblocks < (1ll << 31) && a.n_heads < 65536 && a.batch < 65536, "f3r_attn_fwd: grid too large");
  dim3 grid((unsigned)qblocks, (unsigned)a.n_heads, (unsigned)a.batch);
  hipLaunchKernelGGL((attn_kernel_sp<T, NW, PROF>), grid, dim3(NW * 64), 0, s, a);
  return f3r_check_launch("f3r_attn_fwd");
}
