// MEASUREMENT BUILD ONLY (tools/lab/build_lab.sh -> tools/lab/libf3r_hip_lab.so): the attention variant study of DESIGN.md section 6.
// This is the round-1 kernel source with all of its OPT bits, ablations and per-section timers; the product kernel
// (fast3r_amd/csrc/f3r_attn.hip) is variant 72 of this file with everything else removed.  It replaces f3r_attn.o in the lab
// library and adds two entry points the product does not have: f3r_attn_set_variant / f3r_attn_read_prof (process-global knobs).
//
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

// OPT bit 5 of attn_kernel (measurement builds only): wave 0 of workgroup (0,0,0) accumulates s_memtime deltas of the
// four sections of its tile loop here: [0] QK^T MFMAs, [1] softmax, [2] P V MFMAs, [3] LDS write + barrier, [4] tiles.
__device__ unsigned long long g_attn_prof[8];

constexpr int AT_KB = 64;   // keys per tile
constexpr int AT_TILE = 64 * 64;

__device__ __forceinline__ int aswz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

// NW   waves per workgroup (4 or 8)
// QPW  32-query blocks per wave (1 or 2): with 2, every K / V^T fragment read from LDS feeds two MFMAs
// OPT  bit 0: skip the O rescale when no running max of the wave moved (exact, wave-uniform branch)
//      bit 1: s_setprio 1 around the MFMA clusters
//      bit 2: ABLATION (timing experiments only, wrong results): never reload K/V after the first tile
//      bit 3: ABLATION (timing only): no softmax -- P = S converted to lowp
//      bit 12: no effect on the body: kernel-name tag for the batched (encoder) launches, see attn_dispatch case 55
//      bit 11: pin the K-fragment reads two steps ahead of their MFMAs (sched_group_barrier)
//      bit 10: LAZY reference (needs bit 9): no per-tile max; a lane's 32-key partial row sum >= 64 triggers the rare re-base
//      bit 9: the running max enters the scores through the MFMA itself: a fifth k-step whose K-side fragment is the constant
//             (1, 1, 0, ...) and whose Q-side fragment is (-m_hi, -m_lo, 0, ...) per query (the max kept as an exact sum of two
//             lowp numbers), with Q pre-multiplied by scale*log2(e).  Scores then come out as s' = q.k*c - m and P = exp2(s')
//             needs no per-element subtract/fma: -64 VALU for +4 MFMA per 64 x 64 tile.  Implies bits 0 and 8.
//      bit 8: fewer instructions per tile (the per-wave issue rate is the measured bound): row sums by v_dot2c on the packed P
//             (2 elements per instruction; the sum then is over the ROUNDED probabilities, i.e. exactly what P V multiplies),
//             and pointer-increment addressing of the K / V^T staging loads instead of a 64-bit multiply-add per chunk and tile
//      bit 7: static priority: one s_setprio 1 for the younger half of an 8-wave workgroup (waves 4-7) before the loop
//             (issue arbitration is by priority, then age: MI355X_MICROARCH.md "Two waves per SIMD")
//      bit 6: stage K / V^T tiles with global_load_lds (HBM -> LDS DMA, swizzle applied to the per-lane source address)
//             instead of through registers: no staging VGPRs, no ds_write pass, no vmcnt -> ds_write chain before the barrier
//      bit 4: packed fp32 math (v_pk_fma_f32 / v_pk_add_f32) for the exponent argument and the row sums: a wave issues one
//             instruction per ~5 cycles whatever it is (tools/ubench/valu_rate.hip), so halving the count of these pays
template <class T, int NW, int QPW, int OPT, int MINW>
__global__ __launch_bounds__(NW * 64, MINW) void attn_kernel(const f3r_attn_args p) {
  constexpr int NT = NW * 64;
  constexpr int QB = NW * QPW * 32;  // queries per workgroup
  constexpr bool SPLIT = NT > 512;   // 1024 threads: waves 0-7 stage the K tile, waves 8-15 the V^T tile (one chunk each)
  constexpr int CPT = SPLIT ? 1 : 512 / NT;  // 16-byte chunks of EACH tile staged per thread (of one tile when SPLIT)
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * 2 * AT_TILE];  // [buf][K | Vt][64][64] = 32 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = (OPT & 64) ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);  // scalar for the DMA path's M0 / block math
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
    for (int ds = 0; ds < 4; ++ds) {
      u32x4 raw = *(const u32x4*)(Qg + qrow[qb] * p.ldq + head * 64 + ds * 16 + g * 8);
      if ((OPT & 512) && !p.q_prescaled) {  // the fifth k-step needs the scores in exp2 units
        const float cq = p.scale * 1.44269504088896340736f;
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = pack2<T>(lo_f<T>(raw[j]) * cq, hi_f<T>(raw[j]) * cq);
      }
      qf[qb][ds] = as_vec8<T>(raw);
    }
  }

  // ---- staging role: chunk ids tid + i*NT of the combined [K tile | V^T tile] chunk space (id < 512: K, else V^T);
  // with NT = 1024 a wave stages either K or V^T chunks (wave-uniform), with NT <= 512 every thread stages both kinds
  const int sch = tid & 7;
  const int srow0 = SPLIT ? ((tid >> 3) & 63) : (tid >> 3);

  // flattened (segment, tile) iteration.  The current segment's base pointers live in registers and are re-read from
  // the kernel arguments only when the walk crosses into the next segment (scalar loads stay off the per-tile path).
  int seg_ld = -1;
  int64_t key_ld = 0;  // first key of the next tile to load inside the current segment
  int64_t seg_keys = 0, seg_ldvt = 0;
  const uint16_t* Kg = nullptr;
  const uint16_t* Vg = nullptr;
  const uint16_t* kp[SPLIT ? 1 : 512 / NT];  // OPT bit 8: per-chunk running pointers into the current tile
  const uint16_t* vp[SPLIT ? 1 : 512 / NT];
  const int64_t kstep = (int64_t)AT_KB * p.ldk;
  auto next_segment = [&]() {
    key_ld = 0;
    seg_keys = 0;
    for (++seg_ld; seg_ld < p.n_seg; ++seg_ld)
      if (p.seg_len[seg_ld] > 0) {
        seg_keys = p.seg_len[seg_ld];
        seg_ldvt = p.ldvt[seg_ld];
        Kg = (const uint16_t*)p.k_seg[seg_ld] + (int64_t)b * p.k_batch_stride[seg_ld] + head * 64 + sch * 8;
        Vg = (const uint16_t*)p.vt_seg[seg_ld] + (int64_t)b * p.vt_batch_stride[seg_ld] + (int64_t)head * 64 * seg_ldvt + sch * 8;
        if (OPT & 256) {
#pragma unroll
          for (int i = 0; i < (SPLIT ? 1 : 512 / NT); ++i) {
            const int srow = srow0 + i * (NT / 8);
            kp[i] = Kg + (int64_t)srow * p.ldk;
            vp[i] = Vg + (int64_t)srow * seg_ldvt;
          }
        }
        break;
      }
  };
  next_segment();  // the host guarantees at least one non-empty segment

  u32x4 rk[CPT], rv[SPLIT ? 1 : CPT];  // staged chunks (SPLIT: only rk is used, for either tile)
  int valid_ld = 0;  // valid keys of the tile held in (rk, rv)
  auto load_next = [&]() -> bool {
    if (seg_keys == 0) return false;
    const int64_t rem = seg_keys - key_ld;
    valid_ld = rem < AT_KB ? (int)rem : AT_KB;
    if (!(OPT & 4) || key_ld == 0) {
      if constexpr (SPLIT) {
        u32x4 z = {0u, 0u, 0u, 0u};
        rk[0] = z;
        if (tid < 512) {
          if (srow0 < valid_ld) rk[0] = *(const u32x4*)(Kg + (key_ld + srow0) * p.ldk);
        } else {
          rk[0] = *(const u32x4*)(Vg + (int64_t)srow0 * seg_ldvt + key_ld);
        }
      } else if (OPT & 256) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
          const int srow = srow0 + i * (NT / 8);
          u32x4 z = {0u, 0u, 0u, 0u};
          rk[i] = z;
          if (srow < valid_ld) rk[i] = *(const u32x4*)kp[i];
          rv[i] = *(const u32x4*)vp[i];
          kp[i] += kstep;
          vp[i] += AT_KB;
        }
      } else {
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
    }
    key_ld += AT_KB;
    if (key_ld >= seg_keys) next_segment();
    return true;
  };
  // DMA form of load_next + store_tile (OPT bit 6): every wave-instruction fills 8 rows x 128 B of the K (resp. V^T) image
  // of `buf` straight from HBM/L2 (global_load_lds): no staging registers, no ds_write pass.  Addressing is a wave-uniform
  // tile base (scalar pointer, advanced per tile on the SALU) plus a lane-constant 32-bit byte offset, so the per-tile cost
  // is the DMA instructions themselves; only the ragged last tile of a segment recomputes its K offsets (row clamp).
  constexpr bool DMA = (OPT & 64) != 0;
  constexpr int DPW = NW >= 8 ? 1 : 8 / NW;  // DMA instructions per wave and per tile (8 x 1 KB = one 8 KB tile; DMA bodies use NW <= 8)
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  const int d_lrow = lane >> 3;
  uint32_t dk_off[DPW], dv_off[DPW], d_lch[DPW];
  const char* dKb = nullptr;  // wave-uniform: first K row of the next tile, this head
  const char* dVb = nullptr;  // wave-uniform: V^T row head * 64, first key of the next tile
  auto dma_segment = [&]() {  // after next_segment(): bases and per-lane offsets of the new segment
    if (seg_keys == 0) return;
    dKb = (const char*)((const uint16_t*)p.k_seg[seg_ld] + (int64_t)b * p.k_batch_stride[seg_ld] + head * 64);
    dVb = (const char*)((const uint16_t*)p.vt_seg[seg_ld] + (int64_t)b * p.vt_batch_stride[seg_ld] + (int64_t)head * 64 * seg_ldvt);
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int row = (wid * DPW + i) * 8 + d_lrow;
      d_lch[i] = (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
      dk_off[i] = (uint32_t)row * (uint32_t)p.ldk * 2u + d_lch[i];
      dv_off[i] = (uint32_t)row * (uint32_t)seg_ldvt * 2u + d_lch[i];
    }
  };
  if (DMA) dma_segment();
  uint32_t dv_off_v[DPW];
  const char* dVb_v = nullptr;
  auto dma_next_v = [&](int buf) {
    uint16_t* vt = lds + buf * 2 * AT_TILE + AT_TILE;
#pragma unroll
    for (int i = 0; i < DPW; ++i)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(dVb_v + dv_off_v[i]), (lds_ptr_t)(vt + (wid * DPW + i) * 8 * 64), 16, 0, 0);
  };
  auto dma_next = [&](int buf) -> bool {
    if (seg_keys == 0) return false;
    const int64_t rem = seg_keys - key_ld;
    valid_ld = rem < AT_KB ? (int)rem : AT_KB;
    uint16_t* kt = lds + buf * 2 * AT_TILE;
    uint16_t* vt = kt + AT_TILE;
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int blk = wid * DPW + i;  // 8-row block of the tile (wave-uniform)
      uint32_t ko = dk_off[i];
      if (valid_ld < AT_KB) {  // rows past the segment end: re-read the last valid row (masked in the softmax)
        const int row = blk * 8 + d_lrow;
        const int krow = row < valid_ld ? row : valid_ld - 1;
        ko = (uint32_t)krow * (uint32_t)p.ldk * 2u + d_lch[i];
      }
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(dKb + ko), (lds_ptr_t)(kt + blk * 8 * 64), 16, 0, 0);
      if (!(OPT & 524288)) {
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(dVb + dv_off[i]), (lds_ptr_t)(vt + blk * 8 * 64), 16, 0, 0);
      } else {
        dv_off_v[i] = dv_off[i];  // OPT bit 19: the V^T pieces of this tile are issued later in the iteration (dma_next_v)
      }
    }
    dVb_v = dVb;
    dKb += kstep * 2;
    dVb += AT_KB * 2;
    key_ld += AT_KB;
    if (key_ld >= seg_keys) {
      next_segment();
      dma_segment();
    }
    return true;
  };
  auto store_tile = [&](int buf) {
    uint16_t* kt = lds + buf * 2 * AT_TILE;
    uint16_t* vt = kt + AT_TILE;
    if constexpr (SPLIT) {
      *(u32x4*)((tid < 512 ? kt : vt) + aswz(srow0, sch)) = rk[0];
    } else {
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int srow = srow0 + i * (NT / 8);
        *(u32x4*)(kt + aswz(srow, sch)) = rk[i];
        *(u32x4*)(vt + aswz(srow, sch)) = rv[i];
      }
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
    if (p.state_in) {    // resume an online softmax started by an earlier launch over other K/V segments
      const int64_t row = (int64_t)b * p.tq + qrow[qb];
      const float* so = p.st_o + row * ((int64_t)p.n_heads * 64) + head * 64;
      const float* sm = p.st_ml + (row * p.n_heads + head) * 4;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const float4v v = *(const float4v*)(so + db * 32 + 8 * rq + 4 * g);
          o[qb][db][rq * 4 + 0] = v[0]; o[qb][db][rq * 4 + 1] = v[1]; o[qb][db][rq * 4 + 2] = v[2]; o[qb][db][rq * 4 + 3] = v[3];
        }
      m_run[qb] = sm[0];
      l_run[qb] = sm[1 + g];
    }
  }
  const float c = (p.q_prescaled || (OPT & 512)) ? 1.0f : p.scale * 1.44269504088896340736f;  // exp(x*scale) = exp2(x*c)
  // OPT bit 9: running max as m = mh + ml (both exactly representable in the operand type), fragments of the fifth k-step
  constexpr bool MX = (OPT & 512) != 0;
  constexpr bool LAZY = (OPT & 1024) != 0;  // OPT bit 10: no per-tile max, re-base on a row-sum trigger (needs MX)
  constexpr float AT_REBASE_SUM = 64.f;
  // (the bias step is a 32x32x8 MFMA: 2-register operands, lane (row, g) supplies k-slots 4g..4g+3, so slots 0,1 sit in g == 0)
  u32x2 mfrag[QPW];
  const u32x2 onesfrag = {g == 0 ? pack2<T>(1.0f, 1.0f) : 0u, 0u};
  bool first_tile = true;
#pragma unroll
  for (int qb = 0; qb < QPW; ++qb) {
    mfrag[qb] = u32x2{0u, 0u};
    if (MX) {
      if (p.state_in) {  // resumed state: m_run holds mh + ml; split it again (exact: it was stored as such a sum)
        const float h = from_lp<T>(to_lp<T>(m_run[qb]));
        if (g == 0) mfrag[qb][0] = pack2<T>(-h, -(m_run[qb] - h));
        first_tile = false;
      } else {
        m_run[qb] = 0.f;  // reference max of the bias step; the first tile always re-bases it
      }
    }
  }

  // pi: swap bits 2 and 3 of the key row index fed to the MFMA A operand
  const int krow_pi = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);

  if (DMA) {
    dma_next(0);  // tile 0 always exists
    if (OPT & 524288) dma_next_v(0);
  } else {
    load_next();
  }
  int valid_cur = valid_ld;
  if (!DMA) store_tile(0);
  __syncthreads();
  // Everything loaded so far (Q fragments, tile 0) has landed.  Say so with a waitcnt the compiler models: without it
  // the loop body waits for the loop-invariant Q registers with vmcnt(N) counts that, in steady state, land on the
  // K/V prefetch just issued for the next tile and serialise its latency with the MFMAs of every tile.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched
  int cur = 0;
  bool have = true;
  if ((OPT & 128) && NW == 8) {
    if (__builtin_amdgcn_readfirstlane(tid) >= 256) __builtin_amdgcn_s_setprio(1);
  }
  unsigned long long tq_ = 0, ts_ = 0, tp_ = 0, tb_ = 0, nt_ = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  const unsigned long long t_loop0 = (OPT & 32) ? __builtin_readcyclecounter() : 0ull;
  while (have) {
    const int valid = valid_cur;
    // OPT bits 13-14: WHERE in the iteration the next tile's DMA is issued (its issue cost depends on what else the phase is doing):
    // 0 loop top, 1 after the Q K^T MFMAs, 2 between the two query blocks of the softmax, 3 after the softmax
    constexpr int DMA_AT = DMA ? ((OPT >> 13) & 3) : 0;
    constexpr int PRIO_HI = (OPT & 262144) ? 3 : 1;  // bit 18: priority 3 instead of 1
    // bits 15 / 16: no s_setprio around the P V / Q K^T cluster (bit 1 set)
    bool more = false;
    if (DMA_AT == 0) more = DMA ? dma_next(cur ^ 1) : load_next();  // DMA: buffer cur^1 was released by the barrier that ended iteration t-1
    if (OPT & 32) c0 = __builtin_readcyclecounter();
    const uint16_t* kt = lds + cur * 2 * AT_TILE;
    const uint16_t* vt = kt + AT_TILE;

    // ---- S^T = K Q^T
    float16v s[QPW][2];
    if ((OPT & 2) && !(OPT & 65536)) __builtin_amdgcn_s_setprio(PRIO_HI);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int qb = 0; qb < QPW; ++qb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[qb][kb][i] = 0.f;
        if (MX) s[qb][kb] = T::mfma32k8(onesfrag, mfrag[qb], s[qb][kb]);  // s' starts at -(mh + ml)
      }
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        const typename T::vec8 a = as_vec8<T>(*(const u32x4*)(kt + aswz(kb * 32 + krow_pi, ds * 2 + g)));
#pragma unroll
        for (int qb = 0; qb < QPW; ++qb) s[qb][kb] = T::mfma32(a, qf[qb][ds], s[qb][kb]);
      }
    }
    if (OPT & 2048) {
      // OPT bit 11: pin the K-fragment reads two steps ahead of the MFMAs that consume them.  Left alone, the scheduler
      // keeps ONE fragment register quad here (read, lgkmcnt(0), 2 MFMAs, read, ...): ~100 cycles of LDS latency per
      // 64 cycles of MFMA work, which only the other wave of the SIMD can fill.
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      if (MX) __builtin_amdgcn_sched_group_barrier(0x008, QPW, 0);  // bias steps (one per query block after CSE)
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        if (st + 2 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, QPW, 0);
      }
    }
    if ((OPT & 2) && !(OPT & 65536) && !(OPT & 131072)) __builtin_amdgcn_s_setprio(0);
    if (OPT & 32) { asm volatile("" :: "v"(s[0][0][0]), "v"(s[QPW - 1][1][15])); c1 = __builtin_readcyclecounter(); }
    // register r of block kb is key  kb*32 + 16*(r>>3) + 8*g + (r&7)  of the tile
    if (valid < AT_KB) {
      // compare compile-time key indices against one lane value (valid - 8 g): nothing loop-invariant for the compiler to
      // hoist into 30 registers held across the whole loop for the sake of this once-per-segment block
      const int vg = valid - 8 * g;
#pragma unroll
      for (int qb = 0; qb < QPW; ++qb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kc = kb * 32 + 16 * (r >> 3) + (r & 7);
            if (kc >= vg) s[qb][kb][r] = -1e30f;
          }
    }
    if (DMA_AT == 1) more = dma_next(cur ^ 1);
    if ((OPT & 2) && (OPT & 131072)) __builtin_amdgcn_s_setprio(0);  // bit 17: the DMA is issued before the priority drops
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
    } else if (MX && LAZY) {
      // The reference m only has to keep P = exp2(s - m) inside the operand type's range; softmax is invariant to it.
      // So no per-tile max at all: compute P against the current reference, and only if a lane's 32-key partial sum
      // reaches AT_REBASE_SUM (some P may have grown past 2, none can have passed 64 otherwise) take the rare path that
      // finds the true tile max, moves the reference there and recomputes P.  The first tile always takes it.
#pragma unroll
      for (int qb = 0; qb < QPW; ++qb) {
        if (DMA_AT == 2 && qb == QPW - 1) more = dma_next(cur ^ 1);
        auto probs = [&](const float delta) -> float {
          float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            u32x4 pk;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float p0 = __builtin_amdgcn_exp2f(s[qb][ks >> 1][(ks & 1) * 8 + 2 * j] - delta);
              const float p1 = __builtin_amdgcn_exp2f(s[qb][ks >> 1][(ks & 1) * 8 + 2 * j + 1] - delta);
              pk[j] = pack2<T>(p0, p1);
              if (j & 1) ps1 = T::sum2(pk[j], ps1); else ps0 = T::sum2(pk[j], ps0);
            }
            pf[qb][ks] = as_vec8<T>(pk);
          }
          return ps0 + ps1;
        };
        float psum = 0.f;
        bool redo = first_tile;
        if (!first_tile) {
          psum = probs(0.f);
          redo = __any(psum >= AT_REBASE_SUM);
        }
        if (redo) {
          float mx = s[qb][0][0];
#pragma unroll
          for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[qb][0][r]);
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][1][r]);
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          const float target = m_run[qb] + (first_tile ? mx : fmaxf(mx, 0.f));
          const float nh = from_lp<T>(to_lp<T>(target));
          const float nl = from_lp<T>(to_lp<T>(target - nh));
          const float delta = (nh + nl) - m_run[qb];
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          m_run[qb] = nh + nl;
          if (g == 0) mfrag[qb][0] = pack2<T>(-nh, -nl);
          l_run[qb] *= alpha;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            o[qb][0][i] *= alpha;
            o[qb][1][i] *= alpha;
          }
          psum = probs(delta);
        }
        l_run[qb] += psum;
      }
      first_tile = false;
    } else if (MX) {
#pragma unroll
      for (int qb = 0; qb < QPW; ++qb) {
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
        float mx = fmaxf(fmaxf(m0, s[qb][0][15]), fmaxf(m2, s[qb][1][15]));  // max of s' = max score - (mh + ml)
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (first_tile || __any(mx > 0.f)) {  // wave-uniform and rare after the first tiles: move the reference max
          const float target = m_run[qb] + (first_tile ? mx : fmaxf(mx, 0.f));
          const float nh = from_lp<T>(to_lp<T>(target));
          const float nl = from_lp<T>(to_lp<T>(target - nh));
          const float delta = (nh + nl) - m_run[qb];  // what the reference really moved by
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          m_run[qb] = nh + nl;  // exact: the sum of two operand-type numbers (this is also what a state_out epilogue stores)
          if (g == 0) mfrag[qb][0] = pack2<T>(-nh, -nl);
          l_run[qb] *= alpha;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            s[qb][0][i] -= delta;
            s[qb][1][i] -= delta;
            o[qb][0][i] *= alpha;
            o[qb][1][i] *= alpha;
          }
        }
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          u32x4 pk;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float p0 = __builtin_amdgcn_exp2f(s[qb][ks >> 1][(ks & 1) * 8 + 2 * j]);
            const float p1 = __builtin_amdgcn_exp2f(s[qb][ks >> 1][(ks & 1) * 8 + 2 * j + 1]);
            pk[j] = pack2<T>(p0, p1);
            if (j & 1) ps1 = T::sum2(pk[j], ps1); else ps0 = T::sum2(pk[j], ps0);
          }
          pf[qb][ks] = as_vec8<T>(pk);
        }
        l_run[qb] += ps0 + ps1;
      }
      first_tile = false;
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
      float psum = 0.f, psum2 = 0.f;
      if (OPT & 16) {
        typedef float float2v __attribute__((ext_vector_type(2)));
        const float2v c2 = {c, c}, nmc2 = {-mc, -mc};
        float2v ps0 = {0.f, 0.f}, ps1 = {0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          u32x4 pk;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float2v sv = {s[qb][ks >> 1][(ks & 1) * 8 + 2 * j], s[qb][ks >> 1][(ks & 1) * 8 + 2 * j + 1]};
            const float2v t = __builtin_elementwise_fma(sv, c2, nmc2);
            float2v e;
            e[0] = __builtin_amdgcn_exp2f(t[0]);
            e[1] = __builtin_amdgcn_exp2f(t[1]);
            if (j & 1) ps1 += e; else ps0 += e;
            pk[j] = pack2<T>(e[0], e[1]);
          }
          pf[qb][ks] = as_vec8<T>(pk);
        }
        ps0 += ps1;
        psum = ps0[0] + ps0[1];
      } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        u32x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][ks >> 1][(ks & 1) * 8 + 2 * j], c, -mc));
          const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][ks >> 1][(ks & 1) * 8 + 2 * j + 1], c, -mc));
          pk[j] = pack2<T>(p0, p1);
          if (OPT & 256) {
            if (j & 1) psum2 = T::sum2(pk[j], psum2); else psum = T::sum2(pk[j], psum);
          } else {
            psum += p0 + p1;
          }
        }
        pf[qb][ks] = as_vec8<T>(pk);
      }
      psum += psum2;
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

    if (OPT & 32) { asm volatile("" :: "v"(pf[0][0]), "v"(pf[QPW - 1][3])); c2 = __builtin_readcyclecounter(); }
    if (DMA_AT == 3) more = dma_next(cur ^ 1);
    if (DMA && (OPT & 524288) && more) dma_next_v(cur ^ 1);
    // ---- O^T += V^T P^T
    if ((OPT & 2) && !(OPT & 32768)) __builtin_amdgcn_s_setprio(PRIO_HI);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const typename T::vec8 a = as_vec8<T>(*(const u32x4*)(vt + aswz(db * 32 + lq, ks * 2 + g)));
#pragma unroll
        for (int qb = 0; qb < QPW; ++qb) o[qb][db] = T::mfma32(a, pf[qb][ks], o[qb][db]);
      }
    if ((OPT & 2) && !(OPT & 32768)) __builtin_amdgcn_s_setprio(0);

    if (OPT & 32) { asm volatile("" :: "v"(o[0][0][0]), "v"(o[QPW - 1][1][15])); c3 = __builtin_readcyclecounter(); }
    if (!DMA && more) store_tile(cur ^ 1);
    valid_cur = valid_ld;
    __syncthreads();  // (DMA: the compiler drains vmcnt before the barrier, i.e. the next tile has landed for every wave)
    if (OPT & 32) {
      const unsigned long long c4 = __builtin_readcyclecounter();
      tq_ += c1 - c0; ts_ += c2 - c1; tp_ += c3 - c2; tb_ += c4 - c3; ++nt_;
    }
    cur ^= 1;
    have = more;
  }
  if ((OPT & 32) && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
    g_attn_prof[0] = tq_; g_attn_prof[1] = ts_; g_attn_prof[2] = tp_; g_attn_prof[3] = tb_; g_attn_prof[4] = nt_;
    g_attn_prof[5] = __builtin_readcyclecounter() - t_loop0;  // whole loop (the rest = issuing the next tile's loads)
  }

  // ---- epilogue: either hand the state to the next launch ...
  if (p.state_out) {
#pragma unroll
    for (int qb = 0; qb < QPW; ++qb) {
      if (!q_ok[qb]) continue;
      const int64_t row = (int64_t)b * p.tq + qrow[qb];
      float* so = p.st_o + row * ((int64_t)p.n_heads * 64) + head * 64;
      float* sm = p.st_ml + (row * p.n_heads + head) * 4;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          float4v v = {o[qb][db][rq * 4 + 0], o[qb][db][rq * 4 + 1], o[qb][db][rq * 4 + 2], o[qb][db][rq * 4 + 3]};
          *(float4v*)(so + db * 32 + 8 * rq + 4 * g) = v;
        }
      if (g == 0) sm[0] = m_run[qb];
      sm[1 + g] = l_run[qb];
    }
    return;
  }
  // ---- ... or normalise and store O[q][head*64 + d], d = db*32 + (r&3) + 8*(r>>2) + 4*g
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

#ifdef F3R_ATTN_LAB
#include "f3r_attn_xp.h"   // software-pipelined body (half-tile stages, pinned issue order)
#include "f3r_attn_lab.h"  // experimental bodies (v2 / v3 / v4 / ping-pong): measured, correct, slower -- see DESIGN.md section 6
#endif

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
    // ---- always built: the product (72), its predecessors quoted in DESIGN.md section 6 and the instrumented product (84)
    case 24: return attn_launch<T, 4, 2, 1, 2>(a, s);  // like 3 without setprio
    case 53: return attn_launch<T, 4, 2, 1793, 2>(a, s);  // 51 + lazy reference max (row-sum trigger)
    case 55:  // 53 + LDS-DMA staging (scalar tile base + lane-constant offsets).  OPT bit 12 changes nothing in the body: it only gives
              // the batched (encoder, 1024 keys per sequence) launches their own kernel name, so that a rocprofv3 --stats line
              // of attn_kernel<.., 1857, ..> averages the fusion launches alone (the roofline kernel of bench.py)
      return a.batch > 1 ? attn_launch<T, 4, 2, 1857 + 4096, 2>(a, s) : attn_launch<T, 4, 2, 1857, 2>(a, s);
    case 71:  // 55 with the next tile's DMA issued after the Q K^T MFMAs (+1 %); same kernel-name split as 55
      return a.batch > 1 ? attn_launch<T, 4, 2, 1857 + 8192 + 4096, 2>(a, s) : attn_launch<T, 4, 2, 1857 + 8192, 2>(a, s);
    case 72:  // product: 71 + s_setprio 1 around both MFMA clusters (= variant 70 with the kernel-name split)
      return a.batch > 1 ? attn_launch<T, 4, 2, 1857 + 8192 + 2 + 4096, 2>(a, s) : attn_launch<T, 4, 2, 1857 + 8192 + 2, 2>(a, s);
    case 70: return attn_launch<T, 4, 2, 1857 + 8192 + 2, 2>(a, s);  // 67 + s_setprio around the MFMA clusters
    case 84: return attn_launch<T, 4, 2, 1857 + 8192 + 2 + 32, 2>(a, s);  // 70 + per-section s_memtime instrumentation
#ifdef F3R_ATTN_LAB  // the variant study of DESIGN.md section 6 (~80 more instantiations per operand type, +75 s of compile time):
                     // F3R_EXTRA_FLAGS=-DF3R_ATTN_LAB bash fast3r_amd/csrc/build.sh
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
    case 25: return attn_launch<T, 8, 2, 1, 2>(a, s);  // 8 waves x 64 q (512 q / workgroup, 1 workgroup / CU)
    case 26: return attn_launch<T, 8, 2, 3, 2>(a, s);  // same + setprio
    case 51: return attn_launch<T, 4, 2, 769, 2>(a, s);  // 49 + running max through a fifth MFMA k-step
    case 52: return attn_launch<T, 8, 2, 769, 2>(a, s);  // 50 + the same
    case 54: return attn_launch<T, 8, 2, 1793, 2>(a, s);  // 52 + the same
    case 56: return attn_launch<T, 8, 2, 1857, 2>(a, s);  // 54 + the same
    case 57: return attn_launch<T, 4, 2, 3905, 2>(a, s);  // 55 + K-fragment reads pinned two steps ahead
    case 58: return attn_launch<T, 8, 2, 3905, 2>(a, s);  // 56 + the same
    case 59: return attn_launch<T, 4, 2, 3841, 2>(a, s);  // 53 + the same
    case 60: return attn_launch<T, 8, 2, 3841, 2>(a, s);  // 54 + the same
    case 61: return attn_launch<T, 4, 2, 1889, 2>(a, s);  // 55 + per-section s_memtime instrumentation
    case 62: return attn_launch<T, 8, 2, 1889, 2>(a, s);  // 56 + the same
    case 67: return attn_launch<T, 4, 2, 1857 + 8192, 2>(a, s);   // 55 with the DMA issued after the Q K^T MFMAs
    case 68: return attn_launch<T, 4, 2, 1857 + 16384, 2>(a, s);  // ... between the two query blocks of the softmax
    case 69: return attn_launch<T, 4, 2, 1857 + 24576, 2>(a, s);  // ... after the softmax
    case 73: return attn_launch<T, 4, 2, 1857 + 2, 2>(a, s);          // 55 + s_setprio (DMA at the loop top)
    case 74: return attn_launch<T, 4, 2, 1857 + 16384 + 2, 2>(a, s);  // 68 + s_setprio (DMA between the softmax query blocks)
    case 75: return attn_launch<T, 8, 2, 1857 + 8192 + 2, 2>(a, s);   // 8 waves, DMA after Q K^T, s_setprio
    case 76: return attn_launch<T, 4, 2, 1857 + 24576 + 2, 2>(a, s);  // 69 + s_setprio (DMA after the softmax)
    case 77: return attn_launch<T, 4, 2, 1857 + 8192 + 2 + 32768, 2>(a, s);   // 70 with s_setprio around Q K^T only
    case 78: return attn_launch<T, 4, 2, 1857 + 8192 + 2 + 65536, 2>(a, s);   // 70 with s_setprio around P V only
    case 79: return attn_launch<T, 4, 2, 1857 + 8192 + 2 + 131072, 2>(a, s);  // 70 with the DMA issued before the priority drops
    case 80: return attn_launch<T, 4, 2, 1857 + 8192 + 2 + 262144, 2>(a, s);  // 70 with priority 3
    case 81: return attn_launch<T, 4, 2, 1857 + 8192 + 2 + 2048, 2>(a, s);    // 70 + K fragments pinned two steps ahead
    case 82: return attn_launch<T, 4, 2, 1857 + 8192 + 2 + 524288, 2>(a, s);  // 70 with the V^T pieces issued after the softmax
    case 63: return attn_launch_xp<T, 4, 0, 2>(a, s);  // xp: P V(h-1) / Q K^T(h+1) MFMAs with the softmax of half h in their shadows
    case 64: return attn_launch_xp<T, 4, 1, 2>(a, s);  // 63 + loop timing
    case 65: return attn_launch_xp<T, 4, 0, 1>(a, s);  // xp with one wave per SIMD (512 registers, no spills)
    case 66: return attn_launch_xp<T, 4, 1, 1>(a, s);  // 65 + loop timing
    case 49: return attn_launch<T, 4, 2, 257, 2>(a, s);  // 24 + dot2 row sums + pointer-increment staging
    case 50: return attn_launch<T, 8, 2, 257, 2>(a, s);  // 25 + the same
    case 45: return attn_launch<T, 8, 2, 129, 2>(a, s);  // 25 + static priority for the younger half
    case 46: return attn_launch<T, 8, 2, 193, 2>(a, s);  // 43 (DMA) + static priority
    case 47: return attn_launch<T, 8, 1, 129, 2>(a, s);  // 8 waves x 32 q + static priority
    case 42: return attn_launch<T, 4, 2, 65, 2>(a, s);  // 24 + global_load_lds staging
    case 43: return attn_launch<T, 8, 2, 65, 2>(a, s);  // 25 + global_load_lds staging
    case 44: return attn_launch<T, 4, 2, 97, 2>(a, s);  // 42 + per-section instrumentation
    case 40: return attn_launch<T, 4, 2, 17, 2>(a, s);  // 24 + packed fp32 softmax arithmetic
    case 41: return attn_launch<T, 8, 2, 17, 2>(a, s);  // 25 + packed
    case 34: return attn_launch<T, 4, 2, 33, 2>(a, s);  // variant 24 + per-section s_memtime instrumentation
    case 35: return attn_launch<T, 4, 1, 33, 3>(a, s);  // variant 5 + instrumentation
    case 32: return attn_launch<T, 16, 1, 1, 4>(a, s); // 16 waves x 32 q, 1 workgroup / CU: 4 waves / SIMD, one staged chunk / thread
    case 33: return attn_launch<T, 16, 1, 9, 4>(a, s); // ABLATION of 32: no softmax
    case 29: return attn_launch<T, 4, 1, 1, 4>(a, s);  // 4 waves x 32 q, 4 workgroups / CU (<= 128 VGPR): 4 waves / SIMD
    case 30: return attn_launch<T, 8, 1, 1, 4>(a, s);  // 8 waves x 32 q, 2 workgroups / CU: 4 waves / SIMD
    case 31: return attn_launch<T, 4, 1, 9, 4>(a, s);  // ABLATION of 29: no softmax
    case 48: return attn_launch_sp<T, 8, 3>(a, s);  // 36 + static priority for the younger half
    case 36: return attn_launch_sp<T, 8, 0>(a, s);  // 3-stage software pipeline with pinned issue order, 8 waves x 32 q
    case 37: return attn_launch_sp<T, 4, 0>(a, s);
    case 39: return attn_launch_sp<T, 4, 2>(a, s);  // occupancy experiment: 37 with ONE workgroup (1 wave / SIMD) per CU
    case 38: return attn_launch_sp<T, 8, 1>(a, s);  // 36 + per-part s_memtime instrumentation  // same, 4 waves (2 workgroups / CU)
    case 27: return attn_launch_pp<T, 0>(a, s);  // ping-pong: 2 x 4 waves, matrix phase || softmax phase
    case 28: return attn_launch_pp<T, 8>(a, s);  // ABLATION of 27: no softmax (timing only)
    case 20: return attn_launch_v4<T, 4, 2, 0>(a, s);  // v4 fragment-prefetch body, 4 waves, 2 WG/CU
    case 21: return attn_launch_v4<T, 8, 2, 0>(a, s);  // v4, 8 waves
    case 22: return attn_launch_v4<T, 4, 3, 0>(a, s);  // v4, 4 waves, 3 WG/CU (<=168 VGPR)
    case 23: return attn_launch_v4<T, 4, 2, 8>(a, s);  // ABLATION of 20: no softmax (timing only)
    case 15: return attn_launch_v3<T, 4, 2, 0>(a, s);  // v3 pipelined body, 4 waves, compiler's own interleave
    case 16: return attn_launch_v3<T, 8, 2, 0>(a, s);  // v3, 8 waves
    case 17: return attn_launch_v3<T, 4, 2, 8>(a, s);  // v3, 4 waves, sched_group_barrier pattern 1 MFMA : 1 DS : 8 VALU
    case 18: return attn_launch_v3<T, 8, 2, 8>(a, s);  // v3, 8 waves, same pattern
    case 19: return attn_launch_v3<T, 4, 2, 6>(a, s);  // v3, 4 waves, 1 : 1 : 6
#endif
    default: f3r_set_error("f3r_attn_fwd: variant %d is not in this build (lab variants need -DF3R_ATTN_LAB)", variant); return F3R_ERR_ARG;
  }
}

// Product default (variant "auto"): the lazy-reference body with LDS-DMA staging, 4 waves / 256 queries per workgroup
// (variant 55 = 1.18x variant 24 at 102 400 and at 327 680 keys, the 8-wave form 56 is 1-4 % behind at both; 71 = 55 with the
// next tile's DMA issued after the Q K^T MFMAs instead of at the loop top, another +1 %; 72 = 71 + s_setprio 1 around the two
// MFMA clusters, +4 % -- only in this combination: with the DMA at the loop top the same s_setprio costs 3 %).
constexpr int AT_AUTO = -2;
constexpr int AT_PRODUCT = 72;

int g_variant = -1;  // -1: not initialised yet

int attn_variant(int64_t total_keys) {
  if (g_variant == -1) {
    const char* e = getenv("F3R_ATTN_VARIANT");
    g_variant = e ? atoi(e) : AT_AUTO;
    if (g_variant < 0) g_variant = AT_AUTO;
  }
  (void)total_keys;
  if (g_variant == AT_AUTO) return AT_PRODUCT;
  return g_variant;
}

}  // namespace

extern "C" int f3r_attn_set_variant(int variant);
extern "C" int f3r_attn_read_prof(unsigned long long* out8);

extern "C" int f3r_attn_read_prof(unsigned long long* out8) {
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_attn_prof), 8 * sizeof(unsigned long long)) == hipSuccess ? F3R_OK : F3R_ERR_LAUNCH;
}

extern "C" int f3r_attn_set_variant(int variant) {
  if (variant < -1 || variant > 84) {
    f3r_set_error("f3r_attn_set_variant: unknown variant %d", variant);
    return F3R_ERR_ARG;
  }
  g_variant = variant < 0 ? AT_AUTO : variant;
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
  if (a.state_in || a.state_out) {
    F3R_REQUIRE(a.st_o && a.st_ml && (((uintptr_t)a.st_o) & 15) == 0 && (((uintptr_t)a.st_ml) & 15) == 0, "f3r_attn_fwd: state buffers null/misaligned");
  }
  if (a.tq == 0) return F3R_OK;
  hipStream_t s = (hipStream_t)stream;
  int variant = attn_variant(total);
  const bool product_body = variant == 0 || variant == 3 || variant == 5 || variant == 24 || variant == 25 || (variant >= 49 && variant <= 84);
  if ((a.state_in || a.state_out) && !product_body) variant = AT_PRODUCT;  // only the product body carries state
  return a.dtype == F3R_F16 ? attn_dispatch<F16>(a, s, variant) : attn_dispatch<BF16>(a, s, variant);
}
