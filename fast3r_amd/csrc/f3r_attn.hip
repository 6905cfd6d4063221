// f3r_attn_fwd: O = softmax(scale * Q K^T) V for head_dim 64 on gfx950 -- the fusion transformer's
// multi-view self-attention (all N*P tokens attend to all N*P tokens) and the encoder's per-view
// attention.  Flash style: the T x T score matrix never exists; fp32 online softmax.
//
// Workgroup = 256 threads = 4 waves, 2 workgroups per CU; a wave owns 64 query rows as two 32-query blocks (256 queries per
// workgroup), the workgroup streams the keys in tiles of 64 through a double-buffered LDS image (K tile [64 key][64 d] and V^T tile
// [64 d][64 key], 8 KB each, 16-byte chunks XOR-swizzled by (row >> 1) & 7 so each ds_read_b128 lane group covers the 64 banks once).
// Tiles are staged HBM/L2 -> LDS by LDS-DMA (global_load_lds): a wave-uniform tile base (scalar pointer, advanced per tile on the
// SALU) plus a lane-constant 32-bit byte offset with the swizzle applied to the per-lane SOURCE address -- no staging registers, no
// ds_write pass.  The next tile's DMA is issued right after the Q K^T MFMAs (where its issue cost is smallest); one barrier per tile.
//
// Per wave and tile: S^T = K Q^T as 32(key) x 32(query) v_mfma_f32_32x32x16 blocks ("swapped" QK^T): in the C layout lane l holds
// query column q = l & 31, i.e. every softmax statistic is lane-local apart from one exchange with lane l ^ 32.  The K rows are fed
// to the MFMA through the permutation pi = (swap bits 2 and 3 of the row index): with it, the 8 accumulator registers r = 8h .. 8h+7
// of lane (q, g = l >> 5) are exactly keys 16 ks + 8 g + 0..7 -- the B-operand fragment of the P V product O^T[d][q] += V^T[d][key]
// P^T[key][q].  So P goes from the QK^T accumulators to the PV operand with a pack and no cross-lane traffic, and V^T (written by the
// QKV GEMM epilogue) is read from LDS as ordinary 16-byte A-operand fragments: no transpose anywhere.  Each K / V^T fragment read
// from LDS feeds the MFMAs of both query blocks.
//
// Softmax, built to minimise the instructions a wave issues per tile (the measured bound, docs/history/ (the lab notes of rounds 1 - 4)):
//   * scores come out of the matrix pipe already in exp2 units and already minus the softmax reference m: Q is pre-multiplied by
//     scale*log2(e) (QKV epilogue, f3r_gemm_args.q_scale) and m enters through a fifth MFMA k-step (v_mfma_f32_32x32x8) whose K-side
//     fragment is the constant (1, 1, 0, ..) and whose Q-side fragment is (-m_hi, -m_lo, 0, ..), m kept as an exact sum of two
//     operand-type numbers: P = exp2(s') needs no per-element subtract / fma;
//   * the reference is LAZY: softmax is invariant to m, which only has to keep P inside the operand type's range.  P is computed against
//     the current reference and only when a lane's 32-key partial row sum reaches 64 (some P may have passed 2; none can have passed 64
//     otherwise) does a rare wave-uniform path find the tile max, move the reference, rescale O / l and recompute P.  The first tile
//     always takes it.  No per-tile max, no per-tile rescale;
//   * row sums by v_dot2c on the PACKED P (two elements per instruction): the sum is over the rounded probabilities, i.e. exactly
//     what P V multiplies;
//   * s_setprio 1 around the two MFMA clusters (the wave inside a cluster wins issue arbitration against its SIMD neighbour's softmax).
//
// K/V arrive as segments (f3r_attn_args.k_seg / vt_seg): the single-GPU path uses one segment, the view-sharded multi-GPU path
// passes the local shard plus the all-gathered remote shards; the online-softmax state (m, l, O) can be parked in fp32 between
// launches over different segments (state_in / state_out).
//
// The variant study that led here (90 variants: other workgroup shapes, register staging, software-pipelined bodies, ping-pong,
// ablations, per-section timers) lives in tools/lab/ and is built only into tools/lab/libf3r_hip_lab.so.
#include "f3r_common.h"

namespace {

constexpr int AT_KB = 64;         // keys per tile
constexpr int AT_TILE = 64 * 64;  // elements of a K or V^T tile
constexpr int AT_NW = 4;          // waves per workgroup
constexpr int AT_QPW = 2;         // 32-query blocks per wave
constexpr int AT_QB = AT_NW * AT_QPW * 32;  // queries per workgroup
constexpr int AT_DPW = 8 / AT_NW;           // DMA instructions per wave, operand and tile (8 x 1 KB = one 8 KB tile)
constexpr float AT_REBASE_SUM = 64.f;

__device__ __forceinline__ int aswz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

// BATCHED only tags the kernel name: the encoder launches (many 1024-key sequences) and the fusion launches (one sequence of all
// keys) then show up as two lines in rocprofv3 --stats, and the fusion line is the roofline kernel of bench.py.  Same body.
// CAUSAL (F.scaled_dot_product_attention(..., is_causal=True) of the LlamaDecoder variant) is a separate instantiation: the position
// bookkeeping and the diagonal masks stay out of the non-causal kernel's loop.
template <class T, bool BATCHED, bool CAUSAL>
__global__ __launch_bounds__(AT_NW * 64, 2) void attn_kernel(const f3r_attn_args p) {
  constexpr int QPW = AT_QPW, DPW = AT_DPW;
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * 2 * AT_TILE];  // [buf][K | Vt][64][64] = 32 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: M0 of the DMA and the block math stay on the SALU
  const int lq = lane & 31;
  const int g = lane >> 5;

  const int head = blockIdx.y;
  const int kv_head = p.kv_group > 1 ? head / p.kv_group : head;  // grouped-query attention: kv_group query heads share one K / V head
  const int b = blockIdx.z;
  const int64_t q0 = (int64_t)blockIdx.x * AT_QB + wid * (QPW * 32);

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
      if (!p.q_prescaled) {  // the reference step needs the scores in exp2 units
        const float cq = p.scale * 1.44269504088896340736f;
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[j] = pack2<T>(lo_f<T>(raw[j]) * cq, hi_f<T>(raw[j]) * cq);
      }
      qf[qb][ds] = as_vec8<T>(raw);
    }
  }

  // ---- flattened (segment, tile) walk.  The current segment's bases live in registers and are re-read from the kernel arguments
  // only when the walk crosses into the next segment (scalar loads stay off the per-tile path).
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  int seg_ld = -1;
  int64_t key_ld = 0;  // first key of the next tile to load inside the current segment
  int64_t seg_keys = 0, seg_ldvt = 0;
  const int64_t kstep = (int64_t)AT_KB * p.ldk;
  const int d_lrow = lane >> 3;
  uint32_t dk_off[DPW], dv_off[DPW], d_lch[DPW];
  const char* dKb = nullptr;  // wave-uniform: first K row of the next tile, this head
  const char* dVb = nullptr;  // wave-uniform: V^T row head * 64, first key of the next tile
  int valid_ld = 0;           // valid keys of the tile most recently issued
  int64_t pos_ld = 0;         // causal: global sequence position of its first key
  auto next_segment = [&]() {
    key_ld = 0;
    seg_keys = 0;
    for (++seg_ld; seg_ld < p.n_seg; ++seg_ld)
      if (p.seg_len[seg_ld] > 0) {
        seg_keys = p.seg_len[seg_ld];
        seg_ldvt = p.ldvt[seg_ld];
        dKb = (const char*)((const uint16_t*)p.k_seg[seg_ld] + (int64_t)b * p.k_batch_stride[seg_ld] + kv_head * 64);
        dVb = (const char*)((const uint16_t*)p.vt_seg[seg_ld] + (int64_t)b * p.vt_batch_stride[seg_ld] + (int64_t)kv_head * 64 * seg_ldvt);
#pragma unroll
        for (int i = 0; i < DPW; ++i) {  // wave-instruction i of a tile fills rows (wid*DPW + i)*8 + lane/8; chunk lane%8 <- swizzled source chunk
          const int row = (wid * DPW + i) * 8 + d_lrow;
          d_lch[i] = (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
          dk_off[i] = (uint32_t)row * (uint32_t)p.ldk * 2u + d_lch[i];
          dv_off[i] = (uint32_t)row * (uint32_t)seg_ldvt * 2u + d_lch[i];
        }
        break;
      }
  };
  next_segment();  // the host guarantees at least one non-empty segment
  auto dma_next = [&](int buf) -> bool {
    if (seg_keys == 0) return false;
    const int64_t rem = seg_keys - key_ld;
    valid_ld = rem < AT_KB ? (int)rem : AT_KB;
    if (CAUSAL) pos_ld = p.seg_pos0[seg_ld] + key_ld;
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
      // V^T rows are padded to ldvt (multiple of 64, pad zeroed by the host), so the chunk is always in bounds
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(dVb + dv_off[i]), (lds_ptr_t)(vt + blk * 8 * 64), 16, 0, 0);
    }
    dKb += kstep * 2;
    dVb += AT_KB * 2;
    key_ld += AT_KB;
    if (key_ld >= seg_keys) next_segment();
    return true;
  };

  // ---- running state: O accumulators, reference m = mh + ml (both exactly representable in the operand type), partial row sums
  float16v o[QPW][2];
  float m_run[QPW], l_run[QPW];
  // (the reference step is a 32x32x8 MFMA: 2-register operands, lane (row, g) supplies k-slots 4g..4g+3, so slots 0,1 sit in g == 0)
  u32x2 mfrag[QPW];
  const u32x2 onesfrag = {g == 0 ? pack2<T>(1.0f, 1.0f) : 0u, 0u};
  bool first_tile = true;
#pragma unroll
  for (int qb = 0; qb < QPW; ++qb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[qb][0][i] = 0.f; o[qb][1][i] = 0.f; }
    m_run[qb] = 0.f;  // the first tile always re-bases it
    l_run[qb] = 0.f;  // this lane's partial row sum (its own 32 keys per tile)
    mfrag[qb] = u32x2{0u, 0u};
    if (p.state_in) {  // resume an online softmax started by an earlier launch over other K/V segments
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
      m_run[qb] = sm[0];  // stored as mh + ml: split it again (exact)
      l_run[qb] = sm[1 + g];
      const float h = from_lp<T>(to_lp<T>(m_run[qb]));
      if (g == 0) mfrag[qb][0] = pack2<T>(-h, -(m_run[qb] - h));
      first_tile = false;
    }
  }

  // pi: swap bits 2 and 3 of the key row index fed to the MFMA A operand
  const int krow_pi = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);

  dma_next(0);  // tile 0 always exists
  int valid_cur = valid_ld;
  int64_t pos_cur = pos_ld;
  const int64_t wave_q0 = p.q_pos0 + q0;  // causal: position of this wave's first query
  __syncthreads();
  // Everything loaded so far (Q fragments, tile 0) has landed.  Say so with a waitcnt the compiler models: without it
  // the loop body waits for the loop-invariant Q registers with vmcnt(N) counts that, in steady state, land on the
  // K/V prefetch just issued for the next tile and serialise its latency with the MFMAs of every tile.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched
  int cur = 0;
  bool have = true;
  while (have) {
    const int valid = valid_cur;
    const uint16_t* kt = lds + cur * 2 * AT_TILE;
    const uint16_t* vt = kt + AT_TILE;
    // ---- S^T = K Q^T - m
    float16v s[QPW][2];
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int qb = 0; qb < QPW; ++qb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[qb][kb][i] = 0.f;
        s[qb][kb] = T::mfma32k8(onesfrag, mfrag[qb], s[qb][kb]);  // s' starts at -(mh + ml)
      }
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        const typename T::vec8 a = as_vec8<T>(*(const u32x4*)(kt + aswz(kb * 32 + krow_pi, ds * 2 + g)));
#pragma unroll
        for (int qb = 0; qb < QPW; ++qb) s[qb][kb] = T::mfma32(a, qf[qb][ds], s[qb][kb]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
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
    // causal: keys beyond a row's own position are masked; a tile that lies entirely beyond the wave's rows is masked whole (its P are
    // exactly 0: it costs its MFMAs but changes nothing -- the first tile of a launch always holds a visible key for every row)
    if (CAUSAL && pos_cur + (AT_KB - 1) > wave_q0) {  // the tile reaches past the first row of this wave (wave-uniform)
#pragma unroll
      for (int qb = 0; qb < QPW; ++qb) {
        const int lim = (int)(wave_q0 + qb * 32 + lq - pos_cur) - 8 * g;  // key kc of the tile is visible iff kc + 8 g <= query position - pos_cur
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kc = kb * 32 + 16 * (r >> 3) + (r & 7);
            if (kc > lim) s[qb][kb][r] = -1e30f;
          }
      }
    }
    // the next tile's DMA goes out here: buffer cur^1 was released by the barrier that ended the previous iteration
    const bool more = dma_next(cur ^ 1);

    // ---- online softmax (fp32) against the lazy reference
    typename T::vec8 pf[QPW][4];
#pragma unroll
    for (int qb = 0; qb < QPW; ++qb) {
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
      if (redo) {  // rare, wave-uniform: move the reference to the tile max (never down), rescale O and l, recompute P
        float mx = s[qb][0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[qb][0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
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
          o[qb][0][i] *= alpha;
          o[qb][1][i] *= alpha;
        }
        psum = probs(delta);
      }
      l_run[qb] += psum;
    }
    first_tile = false;

    // ---- O^T += V^T P^T
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const typename T::vec8 a = as_vec8<T>(*(const u32x4*)(vt + aswz(db * 32 + lq, ks * 2 + g)));
#pragma unroll
        for (int qb = 0; qb < QPW; ++qb) o[qb][db] = T::mfma32(a, pf[qb][ks], o[qb][db]);
      }
    __builtin_amdgcn_s_setprio(0);

    valid_cur = valid_ld;
    pos_cur = pos_ld;
    __syncthreads();  // (the compiler drains vmcnt before the barrier, i.e. the next tile has landed for every wave)
    cur ^= 1;
    have = more;
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

template <class T>
int attn_launch(const f3r_attn_args& a, hipStream_t s) {
  const int64_t qblocks = (a.tq + AT_QB - 1) / AT_QB;
  F3R_REQUIRE(qblocks < (1ll << 31) && a.n_heads < 65536 && a.batch < 65536, "f3r_attn_fwd: grid too large");
  dim3 grid((unsigned)qblocks, (unsigned)a.n_heads, (unsigned)a.batch);
  if (a.causal)
    hipLaunchKernelGGL((attn_kernel<T, false, true>), grid, dim3(AT_NW * 64), 0, s, a);
  else if (a.batch > 1)
    hipLaunchKernelGGL((attn_kernel<T, true, false>), grid, dim3(AT_NW * 64), 0, s, a);
  else
    hipLaunchKernelGGL((attn_kernel<T, false, false>), grid, dim3(AT_NW * 64), 0, s, a);
  return f3r_check_launch("f3r_attn_fwd");
}

}  // namespace

// Which kernel f3r_attn_fwd takes for these arguments (reporting only: bench.py names the roofline kernel with it).
extern "C" const char* f3r_attn_kernel_name(const f3r_attn_args* args) {
  if (!args) return "";
  const f3r_attn_args& a = *args;
  const int hd = a.head_dim == 0 ? 64 : a.head_dim;
  const char* why = "";
  if (a.qk_planes == 2) return f3r_attn_asm_eligible(a, 0, &why) ? "f3r_attn_asm_qk3_f16 (hand-scheduled, three products per score block, csrc/asm/attn_gen.py)" : "";
  if (a.qk_planes == 3) return f3r_attn_asm_eligible(a, 0, &why) ? "f3r_attn_asm_qk3f8_f16 (hand-scheduled, three products per score block, the corrections on the fp8 MFMA, csrc/asm/attn_gen.py)" : "";
  if (a.kernel_sel != 1 && f3r_attn_asm_eligible(a, (a.kernel_sel == 2 || hd != 64) ? 0 : F3R_ATTN_ASM_MIN_KEYS, &why)) {
    if (hd == 80) return a.dtype == F3R_F16 ? "f3r_attn_asm_d80_f16 (hand-scheduled, csrc/asm/attn_gen.py)" : "f3r_attn_asm_d80_bf16 (hand-scheduled, csrc/asm/attn_gen.py)";
    if (hd == 128) return a.dtype == F3R_F16 ? "f3r_attn_asm_d128_f16 (hand-scheduled, csrc/asm/attn_gen.py)" : "f3r_attn_asm_d128_bf16 (hand-scheduled, csrc/asm/attn_gen.py)";
    if (f3r_attn_asm_splits_tail(a))
      return a.dtype == F3R_F16 ? "f3r_attn_asm_f16 + f3r_attn_asm_q256_f16 for the last round (hand-scheduled, csrc/asm/attn_gen.py)"
                                : "f3r_attn_asm_bf16 + f3r_attn_asm_q256_bf16 for the last round (hand-scheduled, csrc/asm/attn_gen.py)";
    if (f3r_attn_asm_uses_q256(a))
      return a.dtype == F3R_F16 ? "f3r_attn_asm_q256_f16 (hand-scheduled, 256-query work items, csrc/asm/attn_gen.py)" : "f3r_attn_asm_q256_bf16 (hand-scheduled, 256-query work items, csrc/asm/attn_gen.py)";
    return a.dtype == F3R_F16 ? "f3r_attn_asm_f16 (hand-scheduled, csrc/asm/attn_gen.py)" : "f3r_attn_asm_bf16 (hand-scheduled, csrc/asm/attn_gen.py)";
  }
  if (hd != 64) return "attn_generic_kernel (f3r_attn_generic.hip)";
  return a.causal ? "attn_kernel<causal> (f3r_attn.hip)" : (a.batch > 1 ? "attn_kernel<batched> (f3r_attn.hip)" : "attn_kernel (f3r_attn.hip)");
}

extern "C" int f3r_attn_fwd(const f3r_attn_args* args, f3r_stream_t stream) {
  F3R_REQUIRE(args != nullptr, "f3r_attn_fwd: null args");
  const f3r_attn_args& a = *args;
  F3R_REQUIRE(a.q && a.o, "f3r_attn_fwd: null q/o");
  F3R_REQUIRE(a.dtype == F3R_F16 || a.dtype == F3R_BF16, "f3r_attn_fwd: bad dtype %d", a.dtype);
  F3R_REQUIRE(a.n_heads > 0 && a.batch > 0 && a.tq >= 0, "f3r_attn_fwd: bad sizes");
  F3R_REQUIRE(a.n_seg >= 1 && a.n_seg <= F3R_MAX_SEG, "f3r_attn_fwd: n_seg %d out of range", a.n_seg);
  const int hd = a.head_dim == 0 ? 64 : a.head_dim;
  F3R_REQUIRE(hd >= 16 && hd <= 128 && hd % 16 == 0, "f3r_attn_fwd: head_dim %d (a multiple of 16 up to 128)", a.head_dim);
  F3R_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldo % 4 == 0, "f3r_attn_fwd: ldq/ldk must be multiples of 8, ldo of 4");
  F3R_REQUIRE(a.qk_planes >= 0 && a.qk_planes <= 3, "f3r_attn_fwd: qk_planes %d", a.qk_planes);
  F3R_REQUIRE(a.reserve_cus >= 0, "f3r_attn_fwd: reserve_cus %d", a.reserve_cus);
  const int qkp = a.qk_planes >= 2 ? 2 : 1;   // elements of a q / k row per head = qkp * head_dim
  F3R_REQUIRE(a.ldq >= a.n_heads * hd * qkp && a.ldo >= a.n_heads * hd, "f3r_attn_fwd: row strides < heads*head_dim");
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
    // the LDS-DMA addresses a tile as scalar base + 32-bit lane offset: 64 rows of K / V^T must span < 4 GiB
    F3R_REQUIRE((int64_t)64 * a.ldk * 2 < (1ll << 32) && (int64_t)64 * a.ldvt[s] * 2 < (1ll << 32), "f3r_attn_fwd: ldk / ldvt too large for 32-bit tile offsets");
    total += a.seg_len[s];
  }
  F3R_REQUIRE(total > 0, "f3r_attn_fwd: no keys");
  F3R_REQUIRE(a.kv_group >= 0 && (a.kv_group <= 1 || a.n_heads % a.kv_group == 0), "f3r_attn_fwd: kv_group %d does not divide n_heads %d", a.kv_group, a.n_heads);
  const int kv_heads = a.kv_group > 1 ? a.n_heads / a.kv_group : a.n_heads;
  F3R_REQUIRE(a.ldk >= kv_heads * hd * qkp, "f3r_attn_fwd: ldk < kv heads * head_dim");
  if (a.state_in || a.state_out) {
    F3R_REQUIRE(a.st_o && a.st_ml && (((uintptr_t)a.st_o) & 15) == 0 && (((uintptr_t)a.st_ml) & 15) == 0, "f3r_attn_fwd: state buffers null/misaligned");
  }
  F3R_REQUIRE(a.kernel_sel >= 0 && a.kernel_sel <= 2, "f3r_attn_fwd: kernel_sel %d", a.kernel_sel);
  if (a.tq == 0) return F3R_OK;
  hipStream_t s = (hipStream_t)stream;
  if (a.qk_planes >= 2) {   // hi + lo planes of Q and K: one kernel per layout reads them
    const char* why = "";
    if (a.kernel_sel != 1 && f3r_attn_asm_eligible(a, 0, &why)) return f3r_attn_asm_launch(a, s);
    f3r_set_error("f3r_attn_fwd: qk_planes 2 / 3 but the launch is not eligible for the hand-scheduled three-product kernel: %s", a.kernel_sel == 1 ? "kernel_sel 1" : why);
    return F3R_ERR_UNSUPPORTED;
  }
  if (a.kernel_sel != 1) {  // the hand-scheduled one-wave-per-SIMD kernel where the launch allows it (include/f3r.h)
    const char* why = "";
    if (f3r_attn_asm_eligible(a, (a.kernel_sel == 2 || hd != 64) ? 0 : F3R_ATTN_ASM_MIN_KEYS, &why)) return f3r_attn_asm_launch(a, s);
    if (a.kernel_sel == 2) {
      f3r_set_error("f3r_attn_fwd: kernel_sel 2 (hand-scheduled kernel) but the launch is not eligible: %s", why);
      return F3R_ERR_UNSUPPORTED;
    }
  }
  if (hd != 64) return f3r_attn_generic_launch(a, s);
  return a.dtype == F3R_F16 ? attn_launch<F16>(a, s) : attn_launch<BF16>(a, s);
}
