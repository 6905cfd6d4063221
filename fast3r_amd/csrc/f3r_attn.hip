// f3r_attn_fwd: O = softmax(scale * Q K^T) V for head_dim 64 on gfx950 -- the fusion transformer's
// multi-view self-attention (all N*P tokens attend to all N*P tokens) and the encoder's per-view
// attention.  Flash style: the T x T score matrix never exists; fp32 online softmax.
//
// Workgroup = 512 threads = 8 waves; a wave owns 32 query rows (256 per workgroup), the workgroup
// streams the keys in tiles of 64 through a double-buffered LDS image (K tile [64 key][64 d] and V^T
// tile [64 d][64 key], 8 KB each, 16-byte chunks XOR-swizzled by (row >> 1) & 7 so each ds_read_b128
// lane group covers the 64 banks once).  A tile is fetched global -> registers right before the MFMAs of
// the previous tile and written to the other LDS buffer right after them (one barrier per tile).
//
// Per wave and tile: S^T = K Q^T as two 32(key) x 32(query) v_mfma_f32_32x32x16 blocks ("swapped" QK^T):
// in the C layout lane l holds query column q = l & 31, i.e. every softmax statistic is lane-local apart
// from one exchange with lane l ^ 32.  The K rows are fed to the MFMA through the permutation
// pi = (swap bits 2 and 3 of the row index): with it, the 8 accumulator registers r = 8h .. 8h+7 of
// lane (q, g = l >> 5) are exactly keys 16 ks + 8 g + 0..7 -- the B-operand fragment of the P V product
// O^T[d][q] += V^T[d][key] P^T[key][q].  So P goes from the QK^T accumulators to the PV operand with a
// pack and no cross-lane traffic, and V^T (written by the QKV GEMM epilogue) is read from LDS as
// ordinary 16-byte A-operand fragments: no transpose anywhere.
//
// K/V arrive as segments (f3r_attn_args.k_seg / vt_seg): the single-GPU path uses one segment, the
// view-sharded multi-GPU path passes the local shard plus the all-gathered remote shards.
#include <stdlib.h>

#include <type_traits>

#include "f3r_common.h"

namespace {

constexpr int AT_KB = 64;   // keys per tile
constexpr int AT_TILE = 64 * 64;

__device__ __forceinline__ int aswz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

// NW   waves per workgroup (4 or 8)
// QPW  32-query blocks per wave (1 or 2): with 2, every K / V^T fragment read from LDS feeds two MFMAs
// OPT  bit 0: skip the O rescale when no running max of the wave moved (exact, wave-uniform branch)
//      bit 1: s_setprio 1 around the MFMA clusters
//      bit 2: ABLATION (timing experiments only, wrong results): never reload K/V after the first tile
//      bit 3: ABLATION (timing only): no softmax -- P = S converted to lowp
template <class T, int NW, int QPW, int OPT, int MINW>
__global__ __launch_bounds__(NW * 64, MINW) void attn_kernel(const f3r_attn_args p) {
  constexpr int NT = NW * 64;
  constexpr int QB = NW * QPW * 32;  // queries per workgroup
  constexpr int CPT = 512 / NT;      // 16-byte chunks of each tile staged per thread
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * 2 * AT_TILE];  // [buf][K | Vt][64][64] = 32 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int lq = lane & 31;
  const int g = lane >> 5;

  const int head = blockIdx.y;
  const int b = blockIdx.z;
  const int64_t q0 = (int64_t)blockIdx.x * QB + wid * (QPW * 32);

  // ---- Q fragments (B operand of K Q^T): lane (q, g) holds Q[q][ds*16 + g*8 .. +7]
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
    for (int ds = 0; ds < 4; ++ds)
      qf[qb][ds] = as_vec8<T>(*(const u32x4*)(Qg + qrow[qb] * p.ldq + head * 64 + ds * 16 + g * 8));
  }

  // ---- staging role: CPT 16 B chunks of the K tile and of the V^T tile per thread
  const int sch = tid & 7;
  const int srow0 = tid >> 3;  // + i * (NT / 8)

  // flattened (segment, tile) iteration.  The current segment's base pointers live in registers and are re-read from
  // the kernel arguments only when the walk crosses into the next segment (scalar loads stay off the per-tile path).
  int seg_ld = -1;
  int64_t key_ld = 0;  // first key of the next tile to load inside the current segment
  int64_t seg_keys = 0, seg_ldvt = 0;
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
  next_segment();  // the host guarantees at least one non-empty segment

  u32x4 rk[CPT], rv[CPT];
  int valid_ld = 0;  // valid keys of the tile held in (rk, rv)
  auto load_next = [&]() -> bool {
    if (seg_keys == 0) return false;
    const int64_t rem = seg_keys - key_ld;
    valid_ld = rem < AT_KB ? (int)rem : AT_KB;
    if (!(OPT & 4) || key_ld == 0) {
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int srow = srow0 + i * (NT / 8);
        u32x4 z = {0u, 0u, 0u, 0u};
        rk[i] = z;
        if (srow < valid_ld) rk[i] = *(const u32x4*)(Kg + (key_ld + srow) * p.ldk);
        // V^T rows are padded to ldvt (multiple of 64, pad zeroed by the host), so the chunk is always in bounds
        rv[i] = *(const u32x4*)(Vg + (int64_t)srow * seg_ldvt + key_ld);
      }
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
  float m_run[QPW], l_run[QPW];
#pragma unroll
  for (int qb = 0; qb < QPW; ++qb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[qb][0][i] = 0.f; o[qb][1][i] = 0.f; }
    m_run[qb] = -1e30f;  // running max of the raw scores
    l_run[qb] = 0.f;     // this lane's partial row sum (its own 32 keys per tile)
  }
  const float c = p.q_prescaled ? 1.0f : p.scale * 1.44269504088896340736f;  // exp(x*scale) = exp2(x*c)

  // pi: swap bits 2 and 3 of the key row index fed to the MFMA A operand
  const int krow_pi = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);

  load_next();  // tile 0 always exists
  int valid_cur = valid_ld;
  store_tile(0);
  __syncthreads();
  // Everything loaded so far (Q fragments, tile 0) has landed.  Say so with a waitcnt the compiler models: without it
  // the loop body waits for the loop-invariant Q registers with vmcnt(N) counts that, in steady state, land on the
  // K/V prefetch just issued for the next tile and serialise its latency with the MFMAs of every tile.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched
  int cur = 0;
  bool have = true;
  while (have) {
    const int valid = valid_cur;
    const bool more = load_next();
    const uint16_t* kt = lds + cur * 2 * AT_TILE;
    const uint16_t* vt = kt + AT_TILE;

    // ---- S^T = K Q^T
    float16v s[QPW][2];
    if (OPT & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int qb = 0; qb < QPW; ++qb)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[qb][kb][i] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        const typename T::vec8 a = as_vec8<T>(*(const u32x4*)(kt + aswz(kb * 32 + krow_pi, ds * 2 + g)));
#pragma unroll
        for (int qb = 0; qb < QPW; ++qb) s[qb][kb] = T::mfma32(a, qf[qb][ds], s[qb][kb]);
      }
    }
    if (OPT & 2) __builtin_amdgcn_s_setprio(0);
    // register r of block kb is key  kb*32 + 16*(r>>3) + 8*g + (r&7)  of the tile
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
    // ---- online softmax (fp32)
    typename T::vec8 pf[QPW][4];
    if (OPT & 8) {
#pragma unroll
      for (int qb = 0; qb < QPW; ++qb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          u32x4 pk;
#pragma unroll
          for (int j = 0; j < 4; ++j) pk[j] = pack2<T>(s[qb][ks >> 1][(ks & 1) * 8 + 2 * j], s[qb][ks >> 1][(ks & 1) * 8 + 2 * j + 1]);
          pf[qb][ks] = as_vec8<T>(pk);
        }
    } else
#pragma unroll
    for (int qb = 0; qb < QPW; ++qb) {
      float mx = s[qb][0][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[qb][0][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][1][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[qb], mx);
      const bool moved = m_new > m_run[qb];
      const float mc = m_new * c;
      float psum = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        u32x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][ks >> 1][(ks & 1) * 8 + 2 * j], c, -mc));
          const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][ks >> 1][(ks & 1) * 8 + 2 * j + 1], c, -mc));
          psum += p0 + p1;
          pk[j] = pack2<T>(p0, p1);
        }
        pf[qb][ks] = as_vec8<T>(pk);
      }
      if (!(OPT & 1) || __any(moved)) {
        const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c);
        l_run[qb] = l_run[qb] * alpha + psum;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[qb][0][i] *= alpha; o[qb][1][i] *= alpha; }
      } else {
        l_run[qb] += psum;
      }
      m_run[qb] = m_new;
    }

    // ---- O^T += V^T P^T
    if (OPT & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const typename T::vec8 a = as_vec8<T>(*(const u32x4*)(vt + aswz(db * 32 + lq, ks * 2 + g)));
#pragma unroll
        for (int qb = 0; qb < QPW; ++qb) o[qb][db] = T::mfma32(a, pf[qb][ks], o[qb][db]);
      }
    if (OPT & 2) __builtin_amdgcn_s_setprio(0);

    if (more) store_tile(cur ^ 1);
    valid_cur = valid_ld;
    __syncthreads();
    cur ^= 1;
    have = more;
  }

  // ---- epilogue: normalise, store O[q][head*64 + d], d = db*32 + (r&3) + 8*(r>>2) + 4*g
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

template <class T, int NW, int QPW, int OPT, int MINW>
int attn_launch(const f3r_attn_args& a, hipStream_t s) {
  constexpr int QB = NW * QPW * 32;
  const int64_t qblocks = (a.tq + QB - 1) / QB;
  F3R_REQUIRE(qblocks < (1ll << 31) && a.n_heads < 65536 && a.batch < 65536, "f3r_attn_fwd: grid too large");
  dim3 grid((unsigned)qblocks, (unsigned)a.n_heads, (unsigned)a.batch);
  hipLaunchKernelGGL((attn_kernel<T, NW, QPW, OPT, MINW>), grid, dim3(NW * 64), 0, s, a);
  return f3r_check_launch("f3r_attn_fwd");
}

// kernel variants, selectable with F3R_ATTN_VARIANT for A/B measurement (tools/attn_bench.py)
template <class T>
int attn_dispatch(const f3r_attn_args& a, hipStream_t s, int variant) {
  switch (variant) {
    case 0: return attn_launch<T, 8, 1, 0, 2>(a, s);  // round-1 first light: 8 waves in barrier lockstep
    case 1: return attn_launch<T, 4, 1, 0, 3>(a, s);  // 4-wave workgroups, 3 independent workgroups per CU
    case 2: return attn_launch<T, 4, 1, 3, 3>(a, s);  // + skip-rescale + setprio
    case 3: return attn_launch<T, 4, 2, 3, 2>(a, s);  // 2 query blocks per wave (LDS fragment reuse), 2 WG / CU
    case 4: return attn_launch<T, 8, 1, 3, 2>(a, s);  // 8 waves + skip-rescale + setprio
    case 5: return attn_launch<T, 4, 1, 1, 3>(a, s);  // 4 waves + skip-rescale only
    case 6: return attn_launch<T, 4, 2, 7, 2>(a, s);  // ABLATION of 3: no K/V reloads (timing only)
    case 7: return attn_launch<T, 4, 1, 5, 3>(a, s);  // ABLATION of 5: no K/V reloads (timing only)
    case 12: return attn_launch<T, 4, 2, 11, 2>(a, s);  // ABLATION of 3: no softmax (timing only)
    case 13: return attn_launch<T, 4, 1, 9, 3>(a, s);   // ABLATION of 5: no softmax (timing only)
    case 14: return attn_launch<T, 4, 2, 15, 2>(a, s);  // ABLATION of 3: no softmax, no reloads (timing only)
    case 8: return attn_launch_v2<T, 4, 1, 2>(a, s);   // v2 body, 4 waves x 32 q
    case 9: return attn_launch_v2<T, 4, 2, 2>(a, s);   // v2 body, 4 waves x 64 q
    case 10: return attn_launch_v2<T, 8, 1, 2>(a, s);  // v2 body, 8 waves x 32 q
    case 11: return attn_launch_v2<T, 4, 1, 3>(a, s);  // v2 body, 4 waves x 32 q, 3 workgroups / CU
    case 24: return attn_launch<T, 4, 2, 1, 2>(a, s);  // like 3 without setprio
    case 25: return attn_launch<T, 8, 2, 1, 2>(a, s);  // 8 waves x 64 q (512 q / workgroup, 1 workgroup / CU)
    case 26: return attn_launch<T, 8, 2, 3, 2>(a, s);  // same + setprio
    case 20: return attn_launch_v4<T, 4, 2, 0>(a, s);  // v4 fragment-prefetch body, 4 waves, 2 WG/CU
    case 21: return attn_launch_v4<T, 8, 2, 0>(a, s);  // v4, 8 waves
    case 22: return attn_launch_v4<T, 4, 3, 0>(a, s);  // v4, 4 waves, 3 WG/CU (<=168 VGPR)
    case 23: return attn_launch_v4<T, 4, 2, 8>(a, s);  // ABLATION of 20: no softmax (timing only)
    case 15: return attn_launch_v3<T, 4, 2, 0>(a, s);  // v3 pipelined body, 4 waves, compiler's own interleave
    case 16: return attn_launch_v3<T, 8, 2, 0>(a, s);  // v3, 8 waves
    case 17: return attn_launch_v3<T, 4, 2, 8>(a, s);  // v3, 4 waves, sched_group_barrier pattern 1 MFMA : 1 DS : 8 VALU
    case 18: return attn_launch_v3<T, 8, 2, 8>(a, s);  // v3, 8 waves, same pattern
    case 19: return attn_launch_v3<T, 4, 2, 6>(a, s);  // v3, 4 waves, 1 : 1 : 6
    default: f3r_set_error("f3r_attn_fwd: unknown variant %d", variant); return F3R_ERR_ARG;
  }
}

constexpr int AT_DEFAULT_VARIANT = 24;

int g_variant = -1;

int attn_variant() {
  if (g_variant < 0) {
    const char* e = getenv("F3R_ATTN_VARIANT");
    g_variant = e ? atoi(e) : AT_DEFAULT_VARIANT;
  }
  return g_variant;
}

}  // namespace

extern "C" int f3r_attn_set_variant(int variant) {
  if (variant < -1 || variant > 26) {
    f3r_set_error("f3r_attn_set_variant: unknown variant %d", variant);
    return F3R_ERR_ARG;
  }
  g_variant = variant < 0 ? AT_DEFAULT_VARIANT : variant;
  return F3R_OK;
}

extern "C" int f3r_attn_fwd(const f3r_attn_args* args, f3r_stream_t stream) {
  F3R_REQUIRE(args != nullptr, "f3r_attn_fwd: null args");
  const f3r_attn_args& a = *args;
  F3R_REQUIRE(a.q && a.o, "f3r_attn_fwd: null q/o");
  F3R_REQUIRE(a.dtype == F3R_F16 || a.dtype == F3R_BF16, "f3r_attn_fwd: bad dtype %d", a.dtype);
  F3R_REQUIRE(a.n_heads > 0 && a.batch > 0 && a.tq >= 0, "f3r_attn_fwd: bad sizes");
  F3R_REQUIRE(a.n_seg >= 1 && a.n_seg <= F3R_MAX_SEG, "f3r_attn_fwd: n_seg %d out of range", a.n_seg);
  F3R_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldo % 4 == 0, "f3r_attn_fwd: ldq/ldk must be multiples of 8, ldo of 4");
  F3R_REQUIRE(a.ldq >= a.n_heads * 64 && a.ldk >= a.n_heads * 64 && a.ldo >= a.n_heads * 64, "f3r_attn_fwd: row strides < heads*64");
  F3R_REQUIRE((((uintptr_t)a.q) & 15) == 0 && (((uintptr_t)a.o) & 7) == 0, "f3r_attn_fwd: q/o alignment");
  F3R_REQUIRE(a.q_batch_stride % 8 == 0 && a.o_batch_stride % 4 == 0, "f3r_attn_fwd: batch strides alignment");
  int64_t total = 0;
  for (int s = 0; s < a.n_seg; ++s) {
    F3R_REQUIRE(a.seg_len[s] >= 0, "f3r_attn_fwd: negative segment length");
    if (a.seg_len[s] == 0) continue;
    F3R_REQUIRE(a.k_seg[s] && a.vt_seg[s], "f3r_attn_fwd: null K/V^T segment %d", s);
    F3R_REQUIRE((((uintptr_t)a.k_seg[s]) & 15) == 0 && (((uintptr_t)a.vt_seg[s]) & 15) == 0, "f3r_attn_fwd: K/V^T alignment");
    F3R_REQUIRE(a.ldvt[s] % 64 == 0 && a.ldvt[s] >= a.seg_len[s], "f3r_attn_fwd: ldvt[%d]=%lld must be a multiple of 64 covering the segment (zero padded)", s,
                (long long)a.ldvt[s]);
    F3R_REQUIRE(a.k_batch_stride[s] % 8 == 0 && a.vt_batch_stride[s] % 8 == 0, "f3r_attn_fwd: K/V^T batch stride alignment");
    total += a.seg_len[s];
  }
  F3R_REQUIRE(total > 0, "f3r_attn_fwd: no keys");
  if (a.tq == 0) return F3R_OK;
  hipStream_t s = (hipStream_t)stream;
  const int variant = attn_variant();
  return a.dtype == F3R_F16 ? attn_dispatch<F16>(a, s, variant) : attn_dispatch<BF16>(a, s, variant);
}
