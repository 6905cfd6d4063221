// f3r_gemm256_impl.h: kernel templates of the large-shape path of f3r_gemm (instantiated per operand type by f3r_gemm256.hip [fp16 + host
// logic] and f3r_gemm256_bf16.hip [bf16 + lab variants]: two translation units that compile in parallel) -- out = epilogue(A(M,K) * W(N,K)^T) on a 256 x 256 x 64 tile for gfx950.
//
// One 512-thread workgroup (8 waves, 2 per SIMD) per CU, 128 KiB of LDS: two K-tile buffers of four 16 KiB HALF TILES each
// (A rows 0-127 / 128-255, W rows 0-127 / 128-255; [128 rows][64 k] 16-bit, 128-byte rows, 16-byte chunk c of row r stored at chunk
// c ^ ((r >> 1) & 7): every ds_read_b128 lane group covers the 64 banks exactly once).  Waves are 2 (M) x 4 (N); a wave owns 128 x 64
// outputs = 8 x 4 fragments of v_mfma_f32_16x16x32 (128 fp32 accumulator registers), taken as rows wm*64..+63 of BOTH A half tiles and
// columns wn*32..+31 of BOTH W half tiles: an output quadrant (A half mh, W half nh) then touches exactly one A and one W half tile, every
// half tile is read in ONE phase of the K-tile by all waves, and its LDS slot is free for the next load two phases later.
//
// Schedule (cdna_hip_programming.md "256^2 8-phase", re-derived here because every wait below is placed by counting):
//   * a K-tile is 4 PHASES, one 64 x 32 output quadrant x K = 64 each (16 MFMAs); quadrants (A0,W0) (A1,W0) (A1,W1) (A0,W1) keep
//     both A halves in registers: 12 / 8 / 4 / 0 ds_read_b128 per phase = 24 per 64 MFMAs (0.375 per MFMA);
//   * all global -> LDS traffic is LDS-DMA (global_load_lds, 16 B per lane, no staging registers), ONE half tile (2 instructions per
//     wave) per phase, each issued 5-6 phases (~1.5 K-tiles, ~3000 cycles: HBM latency under load) before the phase that reads it:
//         phase 0 of tile t:  A half 1 of tile t+1        phase 1:  W half 1 of tile t+1
//         phase 2          :  A half 0 of tile t+2        phase 3:  W half 0 of tile t+2
//     with s_waitcnt vmcnt(8) in phases 0, 1 and 3: the DMA queue is never drained in the loop, FOUR half tiles stay in flight across
//     the barriers, and what each wait retires is exactly the half tile the NEXT phase reads;
//   * the two wave rows (wm = 0 / 1, one wave of each per SIMD) run STAGGERED by one barrier: while one does its 16 MFMAs (s_setprio 1)
//     the other issues its ds_reads and LDS-DMA, so the matrix pipe and the LDS / TA pipes alternate owners instead of colliding;
//   * every phase is  [ds_reads, LDS-DMA, (vmcnt)] s_barrier [lgkmcnt(0), 16 MFMA] s_barrier.
// Hazards, with the stagger (a wave of row 1 is one barrier behind a wave of row 0):
//   RAW  LDS-DMA data may be read one phase after the phase whose FIRST barrier follows the issuers' vmcnt wait;
//   WAR  a half tile may be re-staged two phases after the phase that issued its last read.  Reads: A0, W0 in phase 0, A1 in phase 1, W1
//        in phase 2.  Restaged: A0 in phase 2, W0 in phase 3, A1 in phase 0 of the next tile (other buffer: 3 phases), W1 in phase 1.
//
// Operand roles, epilogues, the split-precision K segments and the LDS swizzle are those of f3r_gemm.hip; the implicit-GEMM 3x3
// convolution stages its operand by LDS-DMA too (buffer_load ... lds): out-of-image taps use an out-of-range offset and arrive as zeros.
#pragma once
#include <atomic>

#include "f3r_common.h"
#include "f3r_gemm_epi.h"


namespace {

constexpr int BM = 256, BK = 64, NT = 512;
constexpr int HT = 128 * 64;          // elements of a half tile
// NH = W half tiles per K-tile: 2 -> 256 x 256 outputs, two K-tile buffers (128 KiB); 1 -> 256 x 128 outputs (the 128-channel
// convolutions of the DPT head and every N that is an odd multiple of 128), three K-tile buffers of 3 half tiles (144 KiB)
template <int NH> struct TileCfg {
  static constexpr int BN = 128 * NH;
  static constexpr int BUF = (2 + NH) * HT;          // A_h0 A_h1 W_h0 [W_h1]
  static constexpr int NBUF = NH == 2 ? 2 : 3;
  static constexpr int TAB_ENTRIES = 512;            // CONV: one 16-byte record per K-tile behind the tile buffers (gemm256_body), nk <= 512
  static constexpr int LDS_BYTES = NBUF * BUF * 2 + TAB_ENTRIES * 16;   // 139264 / 155648
};

inline int f3r_num_cus() {  // CUs of the CURRENT device, rounded down to a multiple of 8 (XCDs); cached per device index
  constexpr int MAXD = 64;
  static std::atomic<int> cache[MAXD];  // zero-initialised; a racing first call computes the same value twice
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev >= 0 && dev < MAXD) {
    const int c = cache[dev].load(std::memory_order_relaxed);
    if (c) return c;
  }
  int cu = 0;
  if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 8) cu = 256;
  const int n = cu / 8 * 8;
  if (dev >= 0 && dev < MAXD) cache[dev].store(n, std::memory_order_relaxed);
  return n;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int V>
struct IC {
  static constexpr int value = V;
};

#define F3R_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))   /* vmcnt(n), n <= 15; lgkmcnt / expcnt untouched */
#define F3R_LGKMCNT0() __builtin_amdgcn_s_waitcnt(0xC07F)      /* lgkmcnt(0); vmcnt untouched */

// LAB (tools/lab builds only, -DF3R_GEMM_LAB; 0 in the product): ablation / alternative bits measured by tools/kernel_bench.py --what lab
//   1 no LDS-DMA in the loop   2 no fragment reads in the loop   4 no MFMAs   8 no vmcnt wait   16 no s_setprio
//   32 buffer_load ... lds through a buffer descriptor instead of global_load_lds   64 no sched_barrier pinning of the load section
//   256 / 512 de-phased start: the first round of workgroups starts (wg/8) % 2 resp. % 4 halves / quarters of a tile time late, so the
//   epilogue store bursts of the CUs of an XCD no longer coincide
//   1024 / 2048 (round 6, the conv ablations of f3r_gemm256_f8.hip): no LDS-DMA of the W / of the A operand in the loop
//   128 s_memtime stamps of wave 0 (entry, main loop start, main loop end, epilogue issued, stores retired) -> (uint64*)p.rope_cos [wg][5]
// (1, 2, 4, 8 compute garbage by construction: timing only)
// F8 (F3R_SPLIT_X3F8, CONV3X3 only): the K loop runs on from the nk1 fp16 K-tiles [128 rows][64 k] of A_hi W_hi into 2 x nk1 / 2 fp8 K-tiles
// [128 rows][128 k] -- A_hi8 W_lo8, then A_lo8 W_hi8 -- of the SAME geometry (128-byte rows, same swizzle, same LDS-DMA pieces, same fragment
// reads: a lane's two 16-byte pieces, chunks fg and 4 + fg of its row, are the 32 operand bytes of v_mfma_scale_f32_16x16x128_f8f6f4; both
// operands use the same k positions, which is all the instruction asks for) at the same matrix-pipe time per tile (8 MFMAs of 32 cycles per
// phase instead of 16 of 16): two units of MFMA work instead of X3's three.  Scales: weights one E8M0 byte per output channel and plane
// (w_scale), activations 2^0 (hi8) / 2^-12 (lo8).
// FIN (f3r_gemm_args.fin_w): the epilogue is gemm_epilogue_fin (ReLU -> 1x1 conv to 4 channels -> postprocess), NH == 1 only.
// MERGED (round 6, measurement only): a schedule with half the barriers per MFMA -- see "merged phases" below; measured and not taken.
template <class T, int A_MODE, int EPI, bool SWAP, int STAGGER, int ADDSRC, int NH = 2, int LAB = 0, bool F8 = false, bool FIN = false, bool MERGED = false>
__device__ __forceinline__ void gemm256_body(const f3r_gemm_args& p, uint16_t* smem, const int64_t m0_tile, const int n0_tile, const bool first = true,
                                             const bool has_next = false, const int64_t m0_next = 0, const int n0_next = 0) {
  // PERSISTENT form (gemm256_kernel walks several output tiles per workgroup): `first` = this workgroup's first tile (its opening loads
  // are issued here); otherwise the previous call issued them, between its main loop and its epilogue, so that their latency hides
  // behind the epilogue's stores.  `has_next`: do the same for (m0_next, n0_next) before this tile's epilogue.
  constexpr int BUF = TileCfg<NH>::BUF;
  int64_t m0 = m0_tile;  // the tile the staging lambdas address (switched to the next tile before the epilogue)
  int n0 = n0_tile;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int fr = lane & 15, fg = lane >> 4;
  // PAIRED (lowp-output roles; not QKV, whose RoPE needs columns c and c + 16 in one lane): weight rows are handed to the MFMA so that a lane
  // owns 8 consecutive output columns (GemmFragLayout in f3r_gemm_epi.h).  Fragment f of a wave's 32-row W group then reads rows
  // (i/4)*8 + f*4 + i%4 instead of f*16 + i, and the W half tiles use the swizzle key ((r>>1)&1) | (((r>>3)&3)<<1), which is distinct
  // over exactly those 16 rows x 2 parities (the A half tiles keep (r>>1)&7, distinct over 16 consecutive rows).
  // Kernels whose additive term is fp32 (x + attn(..), x + mlp(..), the image-id rows) read and write fp32 rows: 4 columns are already 16 B
  // there, and the unpaired order keeps a store instruction's 64 B per row contiguous (paired, it would write 16-byte pieces 32 B apart).
  constexpr bool PAIRED = EPI != F3R_EPI_QKV && (ADDSRC == F3R_ADD_NONE || ADDSRC == F3R_ADD_RES_LP);
  static_assert(!(PAIRED && SWAP), "swapped roles exist only in the QKV role");
  auto w_key = [](int r) { return PAIRED ? (((r >> 1) & 1) | (((r >> 3) & 3) << 1)) : ((r >> 1) & 7); };
  uint64_t stamp[5] = {0, 0, 0, 0, 0};
  if (LAB & 128) stamp[0] = __builtin_amdgcn_s_memtime();
  if ((LAB & (256 | 512)) && blockIdx.x < 256) {
    const int PH = (LAB & 512) ? 4 : 2;
    const int phase = (blockIdx.x >> 3) & (PH - 1);
    const int n = phase * (p.Kpad / BK) * (4 / PH);
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(15);
  }

  // ------------------------------------------------------------------ K segments (split precision) and tile counts
  const int nseg = p.split == F3R_SPLIT_NONE ? 1 : (p.split == F3R_SPLIT_W2 ? 2 : 3);
  const int Kpad1 = p.split == F3R_SPLIT_NONE ? p.Kpad : p.Kpad / 2;
  const int nk1 = Kpad1 / BK;
  const int nk8 = nk1 / 2;                       // F8: fp8 K-tiles per correction segment (conv_C % 128 == 0 -> nk1 % 18 == 0)
  const int nk = F8 ? nk1 + 2 * nk8 : nseg * nk1;
  const int ctiles = A_MODE == F3R_A_CONV3X3 ? p.conv_C / 64 : 1;
  const int ctiles8 = A_MODE == F3R_A_CONV3X3 ? p.conv_C / 128 : 1;
  static_assert(!F8 || (A_MODE == F3R_A_CONV3X3 && !SWAP), "fp8 correction segments: convolutions only");
  static_assert(!FIN || (NH == 1 && EPI == F3R_EPI_GENERIC && ADDSRC == 0 && !SWAP), "fused head tail: 256 x 128 tiles, no additive terms");

  // ------------------------------------------------------------------ LDS-DMA source addressing
  // wave w, instruction i of a half tile: rows (w*2 + i)*8 + lane/8, physical chunk lane%8 <- logical chunk (lane%8) ^ ((row>>1)&7)
  uint32_t a_off[2][2], w_off[2][2];  // byte offsets of this lane's 16 B inside the tile's operand panel [half][i]
  uint32_t a_msk[2][2];               // CONV: bit tap = the tap of this lane's pixel lies OUTSIDE the image (or the row is past M)
  const char* const Ab = (const char*)p.A;
  const char* const Alo = (const char*)p.A_lo;
  // CONV: bytes of an operand plane, the range of the buffer descriptors
  const uint32_t a_bytes = A_MODE == F3R_A_CONV3X3 ? (uint32_t)(p.M / ((int64_t)p.conv_OH * p.conv_OW) * p.conv_H * p.conv_W * p.conv_C * 2) : 0u;
  const char* Wb = nullptr;
  uint32_t wsc[NH][2];  // F8: the scale word of this lane's weight row in fragment nf of W half h (byte 0: lo8 plane, byte 1: hi8 plane)
  auto setup_tile = [&](int64_t tm0, int tn0) {
  m0 = tm0;
  n0 = tn0;
  Wb = (const char*)p.W + (int64_t)n0 * p.Kpad * 2;
  if constexpr (F8) {
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        int n = n0 + h * 128 + (wid & 3) * 32 + (PAIRED ? (fr >> 2) * 8 + (fr & 3) + nf * 4 : fr + nf * 16);
        if (n >= p.N) n = p.N - 1;
        wsc[h][nf] = p.w_scale[n];
      }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wid * 2 + i) * 8 + (lane >> 3);
      const int lc = (lane & 7) ^ ((r >> 1) & 7);
      int n = n0 + (h < NH ? h : 0) * 128 + r;
      if (n >= p.N) n = p.N - 1;
      w_off[h][i] = (uint32_t)(((int64_t)(n - n0) * p.Kpad + ((lane & 7) ^ w_key(r)) * 8) * 2);
      int64_t m = m0 + h * 128 + r;
      if (A_MODE == F3R_A_PLAIN) {
        if (m >= p.M) m = p.M - 1;
        a_off[h][i] = (uint32_t)(((m - m0) * p.lda + lc * 8) * 2);
        a_msk[h][i] = 0;
      } else {
        const bool ok = m < p.M;
        const int64_t per_img = (int64_t)p.conv_OH * p.conv_OW;
        const int64_t mm = ok ? m : 0;
        const int b = (int)(mm / per_img);
        const int rem = (int)(mm % per_img);
        const int oy = rem / p.conv_OW, ox = rem - oy * p.conv_OW;
        const int cy = oy * p.conv_stride, cx = ox * p.conv_stride;  // the input pixel under the kernel's centre tap
        a_off[h][i] = (uint32_t)(((((int64_t)b * p.conv_H + cy) * p.conv_W + cx) * p.conv_C + lc * 8) * 2);
        uint32_t msk = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int iy = cy + tap / 3 - 1, ix = cx + tap % 3 - 1;
          if (!(ok && iy >= 0 && iy < p.conv_H && ix >= 0 && ix < p.conv_W)) msk |= 1u << tap;
        }
        a_msk[h][i] = msk;
      }
    }
  };
  setup_tile(m0_tile, n0_tile);

  // cursors: which K-tile the NEXT A / W half-tile pair is loaded for (wave-uniform; clamped at the last tile, see the loop tail)
  // CONV (round 6): where a K-tile's operands live is read from a TABLE in LDS, one 16-byte record per K-tile, filled once per workgroup:
  //   [0] a_delta  byte offset of the tile's 128 operand bytes relative to the lane's own pixel: ((dy W + dx) pix + channel tile * 128), pix = 2 C
  //                bytes in the fp16 planes and in the fp8 planes ([C hi8 | C lo8]: + C for the lo8 half) alike
  //   [1] tap | plane << 4   which bit of the lane's padding mask applies; plane 1 = the operand behind A_lo (X3: the fp16 low plane; F8: the fp8 planes)
  //   [2] w_soff   byte offset of the tile's 128 bytes inside a weight row (plane base + k)
  // The load sections then carry two cursor increments instead of the (segment, tap, channel tile) state machine -- ~35 scalar instructions and
  // five branches per phase.  Why it matters: a wave's load section is an in-order stream of ~75 instructions at 4 - 8 cycles each, longer
  // than the 256 matrix-pipe cycles of the other wave row's MFMA section it is meant to hide behind -- the plain GEMM's phases (fewer
  // instructions, same barriers, same bytes) take 0.38 us, the convolution's took 0.55 (profiles/r05_gemm_small_m_w2_kernel_selection.jsonl vs
  // profiles/r06_conv_x3_vs_x3f8_roles.jsonl).  The record is fetched (one ds_read_b128, broadcast) at the top of the section, ahead of the
  // fragment reads, so its latency hides behind their issue.
  constexpr bool TAB = A_MODE == F3R_A_CONV3X3;
  const int pix = TAB ? p.conv_C * 2 : 0;
  uint32_t* const tab = (uint32_t*)(smem + TileCfg<NH>::NBUF * BUF);
  if (TAB && first) {
    for (int t = tid; t < nk; t += NT) {
      int delta, meta, wso;
      if (!F8 || t < nk1) {
        const int seg = F8 ? 0 : t / nk1, kk = t - seg * nk1, tapn = kk / ctiles, ct = kk - tapn * ctiles;
        delta = ((tapn / 3 - 1) * p.conv_W + (tapn % 3 - 1)) * pix + ct * 128;
        meta = tapn | ((!F8 && seg == 2) ? 16 : 0);
        wso = ((seg == 1 ? Kpad1 : 0) + kk * BK) * 2;
      } else {  // fp8 segments: A_hi8 W_lo8, then A_lo8 W_hi8 (weight rows [2 Kp fp16 | Kp lo8 | Kp hi8] bytes)
        const int u = t - nk1, s8 = u / nk8, kk = u - s8 * nk8, tapn = kk / ctiles8, ct = kk - tapn * ctiles8;
        delta = ((tapn / 3 - 1) * p.conv_W + (tapn % 3 - 1)) * pix + (s8 == 1 ? p.conv_C : 0) + ct * 128;
        meta = tapn | 16;
        wso = (s8 + 2) * Kpad1 + kk * 128;
      }
      *(u32x4*)(tab + t * 4) = u32x4{(uint32_t)delta, (uint32_t)meta, (uint32_t)wso, 0u};
    }
    __syncthreads();
  }
  u32x4 ea = {0u, 0u, 0u, 0u}, ew = ea;  // the records of the A / W cursors' K-tiles (CONV)
  int a_seg = 0, a_kk = 0, a_t = 0;
  int w_seg = 0, w_kk = 0, w_t = 0;
  auto fetch = [&]() {
    if constexpr (TAB) {
      ea = *(const u32x4*)(tab + a_t * 4);
      ew = *(const u32x4*)(tab + w_t * 4);
    }
  };
  bool dry = false;  // advance the cursors without issuing (the loads were issued by the previous tile of this workgroup)
  auto a_advance = [&]() {
    if (a_t + 1 < nk) {
      ++a_t;
      if constexpr (!TAB) {
        ++a_kk;
        if (a_kk == nk1) { a_kk = 0; ++a_seg; }
      }
    }
  };
  auto w_advance = [&]() {
    if (w_t + 1 < nk) {
      ++w_t;
      if constexpr (!TAB) {
        ++w_kk;
        if (w_kk == nk1) { w_kk = 0; ++w_seg; }
      }
    }
  };
  bool in_loop = false;  // LAB only
  auto issue_a = [&](int h, int buf) {  // A half tile h of the cursor's K-tile -> buffer buf
    if (dry) return;
    if ((LAB & (1 | 2048)) && in_loop) return;
    const char* plane = (!TAB && a_seg == 2) ? Alo : Ab;
    uint16_t* dst = smem + buf * BUF + h * HT + wid * 2 * 8 * 64;
    if (A_MODE == F3R_A_PLAIN) {
      const char* base = plane + (m0 * p.lda + (int64_t)a_kk * BK) * 2;
      if (LAB & 32) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(plane + m0 * p.lda * 2), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + i * 8 * 64), 16, (int)a_off[h][i], a_kk * BK * 2, 0, 0);
        return;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + a_off[h][i]), (lds_ptr_t)(dst + i * 8 * 64), 16, 0, 0);
    } else {
      // buffer_load ... lds through a descriptor of the whole plane (below 4 GiB: f3r_gemm256_eligible): lane offset = own pixel + a_delta in
      // 32-bit arithmetic, and a tap outside the image gets the offset 0xFFFFFFFF -- out of range, the hardware writes zeros.  (Until round 6 the
      // loop formed 64-bit lane addresses and pointed padded taps at a zero line: two s_load + an lgkmcnt(0) that also drained the fragment reads,
      // and ~25 more scalar instructions per phase -- the load phases, not the MFMAs, set the tile time: profiles/r06_conv_x3_vs_x3f8_pmc.json.)
      const int a_delta = __builtin_amdgcn_readfirstlane((int)ea[0]);
      const int meta = __builtin_amdgcn_readfirstlane((int)ea[1]);
      const int a_tap = meta & 15;
      const char* pl = (meta & 16) ? Alo : Ab;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)pl, 0, a_bytes, 0x00020000);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int32_t inv = ((int32_t)(a_msk[h][i] << (31 - a_tap))) >> 31;   // v_bfe_i32: -1 = padded tap
        const uint32_t vo = (a_off[h][i] + (uint32_t)a_delta) | (uint32_t)inv;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + i * 8 * 64), 16, (int)vo, 0, 0, 0);
      }
    }
  };
  auto issue_w = [&](int h, int buf) {
    if (dry) return;
    if ((LAB & (1 | 1024)) && in_loop) return;
    uint16_t* dst = smem + buf * BUF + (2 + h) * HT + wid * 2 * 8 * 64;
    if constexpr (TAB) {  // CONV: the tile's offset inside the weight row comes from the table; 32-bit lane offsets through a descriptor of the n-tile's rows
      const int w_soff = __builtin_amdgcn_readfirstlane((int)ew[2]);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + i * 8 * 64), 16, (int)w_off[h][i], w_soff, 0, 0);
      return;
    }
    const char* base = Wb + ((int64_t)(w_seg == 1 ? Kpad1 : 0) + (int64_t)w_kk * BK) * 2;
    if (LAB & 32) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + i * 8 * 64), 16, (int)w_off[h][i], ((w_seg == 1 ? Kpad1 : 0) + w_kk * BK) * 2, 0, 0);
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + w_off[h][i]), (lds_ptr_t)(dst + i * 8 * 64), 16, 0, 0);
  };

  // ------------------------------------------------------------------ fragment read addressing (elements inside a half tile)
  const int sw = (fr >> 1) & 7;
  const int w_row = wn * 32 + (PAIRED ? (fr >> 2) * 8 + (fr & 3) : fr);  // fragment nf adds nf * (PAIRED ? 4 : 16) rows: same key
  int a_rd[2], w_rd[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_rd[ks] = (wm * 64 + fr) * 64 + (((ks * 4 + fg) ^ sw) << 3);
    w_rd[ks] = w_row * 64 + (((ks * 4 + fg) ^ w_key(w_row)) << 3);
  }

  float4v acc[16 * NH];
  typename T::vec8 fa0[2][4], fa1[2][4], fw[2][2];
  // fp8 K-tiles: a fragment is the lane's two 16-byte pieces side by side (8 registers)
  typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  u32x8 ga0[4], ga1[4], gw[2];

  auto read_a = [&](auto e8, auto which, int buf, int mh) {
    if ((LAB & 2) && in_loop) return;
    if constexpr (decltype(e8)::value) {
      auto& g = decltype(which)::value ? ga1 : ga0;
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) {
        const u32x4 c0 = *(const u32x4*)(smem + buf * BUF + mh * HT + a_rd[0] + mf * 16 * 64);
        const u32x4 c1 = *(const u32x4*)(smem + buf * BUF + mh * HT + a_rd[1] + mf * 16 * 64);
        g[mf] = __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    } else {
      auto& f = decltype(which)::value ? fa1 : fa0;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
          f[ks][mf] = as_vec8<T>(*(const u32x4*)(smem + buf * BUF + mh * HT + a_rd[ks] + mf * 16 * 64));
    }
  };
  auto read_w = [&](auto e8, int buf, int nh) {
    if ((LAB & 2) && in_loop) return;
    if constexpr (decltype(e8)::value) {
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        const u32x4 c0 = *(const u32x4*)(smem + buf * BUF + (2 + nh) * HT + w_rd[0] + nf * (PAIRED ? 4 : 16) * 64);
        const u32x4 c1 = *(const u32x4*)(smem + buf * BUF + (2 + nh) * HT + w_rd[1] + nf * (PAIRED ? 4 : 16) * 64);
        gw[nf] = __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
          fw[ks][nf] = as_vec8<T>(*(const u32x4*)(smem + buf * BUF + (2 + nh) * HT + w_rd[ks] + nf * (PAIRED ? 4 : 16) * 64));
    }
  };
  // sa_shift / sb (fp8 tiles): bit offset of the weight plane's scale byte in wsc, and the activations' E8M0 scale byte
  auto mma = [&](auto e8, auto which, int mh, int nh, int sa_shift, int sb) {
    if (LAB & 4) return;
    if constexpr (decltype(e8)::value) {
      auto& g = decltype(which)::value ? ga1 : ga0;
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        const int sa = (int)((wsc[nh < NH ? nh : 0][nf] >> sa_shift) & 0xffu);
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
          const int NF = nh * 2 + nf, MF = mh * 4 + mf;
          acc[NF * 8 + MF] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(__builtin_bit_cast(i32x8, gw[nf]), __builtin_bit_cast(i32x8, g[mf]),
                                                                              acc[NF * 8 + MF], 0, 0, 0, sa, 0, sb);
        }
      }
      // pin the products to THIS phase: nothing but the next phase's MFMAs uses them, and hipcc's code sinking otherwise moves the MFMAs of
      // five phases down to the sixth (all their fragments stay live: 700 bytes of scratch per lane, the matrix pipe idle between the barriers)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) asm volatile("" : "+v"(acc[(nh * 2 + nf) * 8 + mh * 4 + mf]));
    } else {
      auto& f = decltype(which)::value ? fa1 : fa0;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
#pragma unroll
          for (int mf = 0; mf < 4; ++mf) {
            const int NF = nh * 2 + nf, MF = mh * 4 + mf;
            if (SWAP) acc[MF * (2 * NH) + NF] = T::mfma16(f[ks][mf], fw[ks][nf], acc[MF * (2 * NH) + NF]);
            else      acc[NF * 8 + MF] = T::mfma16(fw[ks][nf], f[ks][mf], acc[NF * 8 + MF]);
          }
    }
  };

  // one K-tile out of buffer B (compile-time), 4 phases
#define F3R_PHASE_MMA(WHICH, MH, NH)                     \
  __builtin_amdgcn_sched_barrier(0);                     \
  __builtin_amdgcn_s_barrier();                          \
  F3R_LGKMCNT0();                                        \
  __builtin_amdgcn_sched_barrier(0);                     \
  if (!(LAB & 16)) __builtin_amdgcn_s_setprio(1);        \
  mma(e8, IC<WHICH>{}, MH, NH, sa_shift, sb);            \
  if (!(LAB & 16)) __builtin_amdgcn_s_setprio(0);        \
  __builtin_amdgcn_sched_barrier(0);                     \
  __builtin_amdgcn_s_barrier();
  auto tile = [&](auto bufc, auto e8, int sa_shift, int sb) {
    constexpr int B = decltype(bufc)::value;
    // ---- phase 0: quadrant (A0, W0); A half 1 of the next tile; retire what phase 1 reads (A half 1 of this tile)
    fetch();
    read_w(e8, B, 0);
    if (!(LAB & 64)) __builtin_amdgcn_sched_barrier(0);
    read_a(e8, IC<0>{}, B, 0);
    issue_a(1, B ^ 1);
    a_advance();
    if (!(LAB & 8)) F3R_VMCNT(8);
    F3R_PHASE_MMA(0, 0, 0)
    // ---- phase 1: quadrant (A1, W0); W half 1 of the next tile; retire W half 1 of this tile (phase 2 reads it)
    fetch();
    read_a(e8, IC<1>{}, B, 1);
    issue_w(1, B ^ 1);
    w_advance();
    if (!(LAB & 8)) F3R_VMCNT(8);
    F3R_PHASE_MMA(1, 1, 0)
    // ---- phase 2: quadrant (A1, W1); A half 0 two tiles ahead (its slot was last read in phase 0)
    fetch();
    read_w(e8, B, 1);
    issue_a(0, B);
    F3R_PHASE_MMA(1, 1, 1)
    // ---- phase 3: quadrant (A0, W1); W half 0 two tiles ahead; retire A half 0 and W half 0 of the next tile (its phase 0 reads them)
    fetch();
    issue_w(0, B);
    if (!(LAB & 8)) F3R_VMCNT(8);
    F3R_PHASE_MMA(0, 0, 1)
  };
  // NH == 1 (256 x 128 outputs): a K-tile is 2 phases -- quadrants (A0, W) and (A1, W), 16 MFMAs each -- and 3 half tiles; three
  // buffers, tile t in buffer t % 3.  Reads: A0 and W in phase 0, A1 in phase 1; a slot is restaged two phases after its read:
  //     phase 0 of tile t:  A half 0 + W of tile t+2, then vmcnt(10) (retires A half 1 of tile t, read in phase 1)
  //     phase 1          :  A half 1 of tile t+2,     then vmcnt(8)  (retires A half 0 + W of tile t+1, read in its phase 0)
  // i.e. every half tile is issued 3 phases before the wait that retires it.
  auto tile1 = [&](auto bufc, auto e8, int sa_shift, int sb) {
    constexpr int B = decltype(bufc)::value;
    constexpr int B2 = (B + 2) % 3;
    fetch();
    read_w(e8, B, 0);
    if (!(LAB & 64)) __builtin_amdgcn_sched_barrier(0);
    read_a(e8, IC<0>{}, B, 0);
    issue_a(0, B2);
    issue_w(0, B2);
    w_advance();
    if (!(LAB & 8)) F3R_VMCNT(10);
    F3R_PHASE_MMA(0, 0, 0)
    read_a(e8, IC<1>{}, B, 1);
    issue_a(1, B2);
    a_advance();
    if (!(LAB & 8)) F3R_VMCNT(8);
    F3R_PHASE_MMA(1, 1, 0)
  };
  // ---- merged phases (MERGED; measurement only, NOT the product's schedule): a phase is TWO quadrants = 32 MFMAs between one pair of barriers
  // (the quadrant pairs (A0,W0)+(A1,W0) and (A1,W1)+(A0,W1) need no register the four-phase form does not hold already).  The question it
  // answers: a phase takes ~0.55 us whether it carries 16 KiB or 24 KiB of operands, against 2 x 256 matrix-pipe cycles ~ 0.25 us
  // (profiles/r06_conv_x3_vs_x3f8_pmc.json) -- is the barrier pair the fixed cost?  No: with half the barriers per MFMA the 256 x 128 tile got
  // 9 - 11 % SLOWER and the 256 x 256 tile stayed where it was, with the LDS-DMA issued behind the MFMAs of the phase (formally hazard-free) or
  // in the section that carries the fragment reads (below; the restaged slots' last reads are then only ONE barrier older, which the in-order
  // LDS queue makes safe in practice but no wait proves) -- profiles/r06_conv_merged_phases_*_rejected.jsonl.  What the merged form loses is
  // lead time: with three 48 KiB buffers a one-phase tile must restage the buffer it read one phase ago, so a piece has one phase to land
  // instead of three.
  // NH == 2, tile t in buffer B = t % 2:
  //   P0  reads W0, A0, A1 of B;  issue W1 of t+1 -> B^1;          wait vmcnt(8): W1 of tile t has landed;        MFMA (A0,W0) (A1,W0)
  //   P1  reads W1 of B;          issue A0, W0, A1 of t+2 -> B;    wait vmcnt(8): A0, W0, A1 of t+1 have landed;  MFMA (A1,W1) (A0,W1)
  // NH == 1, tile t in buffer t % 3, one phase per tile:
  //   reads W, A0, A1;  issue tile t+2 -> (t+2) % 3;  wait vmcnt(6): tile t+1 has landed;  MFMA (A0,W) (A1,W)
#define F3R_MMA2_OPEN()                                  \
  __builtin_amdgcn_sched_barrier(0);                     \
  __builtin_amdgcn_s_barrier();                          \
  F3R_LGKMCNT0();                                        \
  __builtin_amdgcn_sched_barrier(0);                     \
  __builtin_amdgcn_s_setprio(1);
#define F3R_MMA2_ISSUE()                                 \
  __builtin_amdgcn_s_setprio(0);                         \
  __builtin_amdgcn_sched_barrier(0);
#define F3R_MMA2_CLOSE()                                 \
  __builtin_amdgcn_sched_barrier(0);                     \
  __builtin_amdgcn_s_barrier();
  auto tileM = [&](auto bufc, auto e8, int sa_shift, int sb) {
    constexpr int B = decltype(bufc)::value;
    fetch();
    read_w(e8, B, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(e8, IC<0>{}, B, 0);
    read_a(e8, IC<1>{}, B, 1);
    issue_w(1, B ^ 1);
    w_advance();
    F3R_VMCNT(8);
    F3R_MMA2_OPEN()
    mma(e8, IC<0>{}, 0, 0, sa_shift, sb);
    mma(e8, IC<1>{}, 1, 0, sa_shift, sb);
    F3R_MMA2_ISSUE()
    F3R_MMA2_CLOSE()
    fetch();
    read_w(e8, B, 1);
    issue_a(0, B);
    issue_w(0, B);
    issue_a(1, B);
    a_advance();
    F3R_VMCNT(8);
    F3R_MMA2_OPEN()
    mma(e8, IC<1>{}, 1, 1, sa_shift, sb);
    mma(e8, IC<0>{}, 0, 1, sa_shift, sb);
    F3R_MMA2_ISSUE()
    F3R_MMA2_CLOSE()
  };
  auto tile1M = [&](auto bufc, auto e8, int sa_shift, int sb) {
    constexpr int B = decltype(bufc)::value;
    constexpr int B2 = (B + 2) % 3;
    fetch();
    read_w(e8, B, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(e8, IC<0>{}, B, 0);
    read_a(e8, IC<1>{}, B, 1);
    issue_a(0, B2);
    issue_w(0, B2);
    w_advance();
    issue_a(1, B2);
    a_advance();
    F3R_VMCNT(6);
    F3R_MMA2_OPEN()
    mma(e8, IC<0>{}, 0, 0, sa_shift, sb);
    mma(e8, IC<1>{}, 1, 0, sa_shift, sb);
    F3R_MMA2_ISSUE()
    F3R_MMA2_CLOSE()
  };
#undef F3R_MMA2_OPEN
#undef F3R_MMA2_ISSUE
#undef F3R_MMA2_CLOSE
#undef F3R_PHASE_MMA

  // ------------------------------------------------------------------ prologue
  // The bias and the additive epilogue terms (fp32 / lowp residuals, image-id rows) are loaded FIRST, straight into the accumulators:
  // they land under the latency of the first tiles (f3r_gemm_epi.h); the compiler's own wait covers their first use.
  typedef GemmFragLayout<2 * NH, 8, 2, 4, PAIRED> Frag;  // 2 fragments from each W half, 4 fragments from each A half
  const int64_t m_base = m0_tile + wm * 64;
  const int n_base = n0_tile + wn * 32;
  auto opening_loads = [&]() {
    a_seg = a_kk = a_t = 0;
    w_seg = w_kk = w_t = 0;
    // (CONV: every issue reads its cursor's table record -- fetch() after each advance; `dry` only moves the cursors)
    if constexpr (NH == 2) {  // all of tile 0 and the first halves of tile 1
      if (!dry) fetch();
      issue_a(0, 0);
      issue_w(0, 0);
      issue_a(1, 0);
      a_advance();
      issue_w(1, 0);
      w_advance();
      if (!dry) fetch();
      issue_a(0, 1);
      issue_w(0, 1);
      if constexpr (MERGED) {  // ... and its A half 1 (the merged schedule issues tile t+2's first three halves in phase P1 of tile t)
        issue_a(1, 1);
        a_advance();
      }
    } else {                  // tiles 0 and 1
      if (!dry) fetch();
      issue_a(0, 0);
      issue_w(0, 0);
      w_advance();
      issue_a(1, 0);
      a_advance();
      if (!dry) fetch();
      issue_a(0, 1);
      issue_w(0, 1);
      w_advance();
      issue_a(1, 1);
      a_advance();
    }
  };
  dry = !first;
  opening_loads();
  dry = false;
  gemm_acc_init_additive<T, Frag, ADDSRC, SWAP>(p, acc, m_base, n_base, lane);
  if (first && MERGED) {
    if constexpr (NH == 2) F3R_VMCNT(8);  // A0, W0, A1 of tile 0 have landed (younger: W1 of tile 0, A0, W0, A1 of tile 1)
    else F3R_VMCNT(6);                    // tile 0 has landed (younger: tile 1)
  } else if (first) {
    F3R_VMCNT(8);  // A half 0 and W half 0 (NH == 1: W) of tile 0 have landed; the four younger half tiles stay in flight
  } else {
    F3R_VMCNT(0);  // the previous tile's epilogue stores are in the queue behind the opening loads: counted waits resume once it is empty
  }
  __builtin_amdgcn_s_barrier();
  if (STAGGER && wm == 1) __builtin_amdgcn_s_barrier();
  if (LAB & 128) stamp[1] = __builtin_amdgcn_s_memtime();

  typedef IC<0> E16;
  typedef IC<1> E8;
  auto T2 = [&](auto bufc, auto e8, int sa_shift, int sb) {
    if constexpr (MERGED) tileM(bufc, e8, sa_shift, sb);
    else tile(bufc, e8, sa_shift, sb);
  };
  auto T1 = [&](auto bufc, auto e8, int sa_shift, int sb) {
    if constexpr (MERGED) tile1M(bufc, e8, sa_shift, sb);
    else tile1(bufc, e8, sa_shift, sb);
  };
  const int nk16 = F8 ? nk1 : nk;  // 16-bit K-tiles; F8: followed by nk1 fp8 K-tiles (nk1 % 6 == 0: both loops start at buffer 0)
  if constexpr (NH == 2) {
    if (LAB) {  // the first pair of tiles loads real fragments, the rest run with the ablated sections
      tile(IC<0>{}, E16{}, 0, 0);
      if (nk > 1) tile(IC<1>{}, E16{}, 0, 0);
      in_loop = true;
    }
    for (int t = LAB ? 2 : 0; t < nk16; t += 2) {
      T2(IC<0>{}, E16{}, 0, 0);
      if (t + 1 < nk16) T2(IC<1>{}, E16{}, 0, 0);
    }
    if constexpr (F8) {  // segment 1 (tiles < nk8): A_hi8 W_lo8, scales (byte 0, 2^0); segment 2: A_lo8 W_hi8, scales (byte 1, 2^-12)
      for (int t = 0; t < nk1; t += 2) {
        T2(IC<0>{}, E8{}, t >= nk8 ? 8 : 0, t >= nk8 ? 115 : 127);
        T2(IC<1>{}, E8{}, t + 1 >= nk8 ? 8 : 0, t + 1 >= nk8 ? 115 : 127);
      }
    }
  } else {
    if (LAB) in_loop = true;  // (256 x 128 tile ablations: every tile runs with the ablated sections -- timing only, the results are garbage)
    for (int t = 0; t < nk16; t += 3) {
      T1(IC<0>{}, E16{}, 0, 0);
      if (t + 1 < nk16) T1(IC<1>{}, E16{}, 0, 0);
      if (t + 2 < nk16) T1(IC<2>{}, E16{}, 0, 0);
    }
    if constexpr (F8) {
      for (int t = 0; t < nk1; t += 3) {
        T1(IC<0>{}, E8{}, t >= nk8 ? 8 : 0, t >= nk8 ? 115 : 127);
        T1(IC<1>{}, E8{}, t + 1 >= nk8 ? 8 : 0, t + 1 >= nk8 ? 115 : 127);
        T1(IC<2>{}, E8{}, t + 2 >= nk8 ? 8 : 0, t + 2 >= nk8 ? 115 : 127);
      }
    }
  }
  // Past the last tile the cursors stay clamped, so the tail re-loads the last tile into half tiles nobody reads any more; drain them
  // before the workgroup's LDS can be handed to the next one.
  F3R_VMCNT(0);
  if (STAGGER && wm == 0) __builtin_amdgcn_s_barrier();
  if (LAB & 128) stamp[2] = __builtin_amdgcn_s_memtime();
  if (has_next) {  // every wave is past its last LDS read and the queue is empty: stage the next tile's opening half tiles under the epilogue
    setup_tile(m0_next, n0_next);
    opening_loads();
  }

  // ------------------------------------------------------------------ epilogue
  if constexpr (FIN) {
    // every wave is past its own vmcnt(0); the barrier makes that true for ALL waves before K-tile buffer 2 (never a target of the next tile's
    // opening loads, which use buffers 0 and 1) becomes the scratch of the cross-wave reduction
    __builtin_amdgcn_s_barrier();
    gemm_epilogue_fin<T, Frag>(p, acc, m0_tile, wm, wn, lane, tid, (float*)(smem + 2 * BUF));
  } else if (SWAP) gemm_epilogue_vt<T, Frag, false>(p, acc, m_base, n_base, lane);
  else gemm_epilogue_default<T, EPI, Frag, false>(p, acc, m_base, n_base, lane);
  if (LAB & 128) {
    __builtin_amdgcn_sched_barrier(0);
    stamp[3] = __builtin_amdgcn_s_memtime();
    F3R_VMCNT(0);
    stamp[4] = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      uint64_t* d = (uint64_t*)p.rope_cos + (int64_t)blockIdx.x * 5;
      for (int i = 0; i < 5; ++i) d[i] = stamp[i];
    }
  }
}

template <class T, int A_MODE, int EPI, int STAGGER, int ADDSRC, int NH, bool F8 = false, bool FIN = false, bool MERGED = false, int LAB = 0>
__global__ __launch_bounds__(NT, 1) void gemm256_kernel(const f3r_gemm_args p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  constexpr int BN = TileCfg<NH>::BN;
  // XCD-aware tile order: consecutive workgroup ids go round-robin over the 8 XCDs; give each XCD a contiguous run of tiles, and
  // inside the run walk GM m-tiles x all n-tiles with m fastest, so the ~32 tiles an XCD runs at once share 8 A panels and all of W
  // through its private 4 MiB L2.
  const int n_tiles_n = (p.N + BN - 1) / BN;
  const int64_t n_tiles_m = (p.M + BM - 1) / BM;
  const int64_t n_wg = n_tiles_m * n_tiles_n;
  auto tile_of = [&](int64_t wg, int64_t& m0, int& n0) {
    {
      const int64_t q = n_wg / 8, r = n_wg % 8;
      const int64_t xcd = wg % 8, idx = wg / 8;
      wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GM = 8;
    const int64_t per_group = (int64_t)GM * n_tiles_n;
    const int64_t grp = wg / per_group;
    const int64_t first_m = grp * GM;
    const int gm = (int)((n_tiles_m - first_m) < GM ? (n_tiles_m - first_m) : GM);
    const int64_t rem_ = wg - grp * per_group;
    const int tn = (int)(rem_ / gm);
    const int64_t tm = first_m + rem_ % gm;
    m0 = tm * BM;
    n0 = tn * BN;
  };
  // PERSISTENT: gridDim.x = min(tiles, CUs) workgroups (a multiple of 8 whenever there is more than one round, so a workgroup's tiles
  // stay on its XCD's contiguous run); workgroup b computes tiles b, b + gridDim.x, ...; between two of them the next tile's opening
  // LDS-DMA loads are issued before the finished tile's epilogue stores (gemm256_body).
  int64_t m0, m0n = 0;
  int n0, n0n = 0;
  bool first = true;
  for (int64_t v = blockIdx.x; v < n_wg; v += gridDim.x) {
    tile_of(v, m0, n0);
    const bool has_next = v + gridDim.x < n_wg;
    if (has_next) tile_of(v + gridDim.x, m0n, n0n);
    bool done = false;
    if constexpr (EPI == F3R_EPI_QKV) {
      if (n0 >= (p.qkv_dq ? p.qkv_dq + (p.N - p.qkv_dq) / 2 : 2 * (p.N / 3))) {  // the V part: swapped operand roles, V^T epilogue
        gemm256_body<T, A_MODE, EPI, true, STAGGER, F3R_ADD_NONE, NH>(p, smem, m0, n0, first, has_next, m0n, n0n);
        done = true;
      }
    }
    if (!done) gemm256_body<T, A_MODE, EPI, false, STAGGER, ADDSRC, NH, LAB, F8, FIN, MERGED>(p, smem, m0, n0, first, has_next, m0n, n0n);
    first = false;
  }
}

template <class T, int A_MODE, int EPI, int STAGGER, int ADDSRC, int NH, bool F8 = false, bool FIN = false, bool MERGED = false, int LAB = 0>
int launch256(const f3r_gemm_args& a, hipStream_t stream) {
  static bool attr_set = false;  // benign race: idempotent
  auto kern = gemm256_kernel<T, A_MODE, EPI, STAGGER, ADDSRC, NH, F8, FIN, MERGED, LAB>;
  constexpr int LDS_BYTES = TileCfg<NH>::LDS_BYTES, BN = TileCfg<NH>::BN;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  const int64_t tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  // one resident workgroup per CU (128 / 144 KiB of LDS): a persistent grid of at most that many workgroups; kernel_sel 5 asks for the
  // one-tile-per-workgroup form (measurements)
  const int64_t grid = (a.kernel_sel == 5 || tiles <= f3r_num_cus()) ? tiles : f3r_num_cus();
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), LDS_BYTES, stream, a);
  return f3r_check_launch("f3r_gemm(256)");
}

// 256-wide tiles when N fills them; 128-wide tiles when N is an odd multiple of 128 (no half-empty tile) -- QKV needs the wide tile
// Which tile form fills the 256 CUs best.  score = (fraction of the last round's slots that do work, over all rounds) x (relative
// main-loop efficiency of the form: 256x256 1.0, 256x128 0.82, 128x128 with two workgroups per CU 0.72 -- tools/kernel_bench.py
// --what gemmsmall / gemm, profiles/r02_kernel_microbench_gemm_small.jsonl).  E.g. 8 views (M = 8192, N = 1024): 128 tiles of 256^2 = half a
// round (0.50), 256 tiles of 256x128 = one full round (0.82), 512 tiles of 128^2 = one full round (0.72) -> 256x128.
inline float tile_score(int64_t tiles, int slots, float eff) {
  if (tiles <= 0) return 0.f;
  const int64_t rounds = (tiles + slots - 1) / slots;
  return eff * (float)tiles / (float)(rounds * slots);
}
inline float score_256(const f3r_gemm_args& a, int nh) {
  return tile_score(((a.M + 255) / 256) * ((a.N + 128 * nh - 1) / (128 * nh)), 256, nh == 2 ? 1.0f : 0.82f);
}
inline float score_128(const f3r_gemm_args& a) { return tile_score(((a.M + 127) / 128) * ((a.N + 127) / 128), 512, 0.72f); }

inline int tile_halves(const f3r_gemm_args& a) {
  if (a.epi == F3R_EPI_QKV) return 2;
  if (a.N % 256 != 0 || a.kernel_sel == 4) return 1;
  if (a.kernel_sel == 2 || a.kernel_sel == 3 || a.kernel_sel == 5) return 2;
  return score_256(a, 1) > score_256(a, 2) ? 1 : 2;
}

template <class T>
int dispatch256(const f3r_gemm_args& a, hipStream_t stream, int stagger) {
#define F3R_L256H(AM, EP, AD, NHV) (stagger ? launch256<T, AM, EP, 1, AD, NHV>(a, stream) : launch256<T, AM, EP, 0, AD, NHV>(a, stream))
#define F3R_L256(AM, EP, AD) (nh == 2 ? F3R_L256H(AM, EP, AD, 2) : F3R_L256H(AM, EP, AD, 1))
  const int add = gemm_additive_pattern(a);
  const int nh = tile_halves(a);
  if (a.a_mode == F3R_A_CONV3X3) return add == F3R_ADD_RES_LP ? F3R_L256(F3R_A_CONV3X3, F3R_EPI_GENERIC, F3R_ADD_RES_LP) : F3R_L256(F3R_A_CONV3X3, F3R_EPI_GENERIC, F3R_ADD_NONE);
  switch (a.epi) {
    case F3R_EPI_GENERIC:
      return add == F3R_ADD_RES_F32 ? F3R_L256(F3R_A_PLAIN, F3R_EPI_GENERIC, F3R_ADD_RES_F32)
           : add == F3R_ADD_ROWADD ? F3R_L256(F3R_A_PLAIN, F3R_EPI_GENERIC, F3R_ADD_ROWADD) : F3R_L256(F3R_A_PLAIN, F3R_EPI_GENERIC, F3R_ADD_NONE);
    case F3R_EPI_QKV: return F3R_L256H(F3R_A_PLAIN, F3R_EPI_QKV, F3R_ADD_NONE, 2);
    default: return F3R_L256(F3R_A_PLAIN, F3R_EPI_CONVT, F3R_ADD_NONE);
  }
#undef F3R_L256
#undef F3R_L256H
}

#ifdef F3R_GEMM_LAB
template <class T, int LAB>
__global__ __launch_bounds__(NT, 1) void gemm256_lab_kernel(const f3r_gemm_args p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  constexpr int BN = 256;
  const int n_tiles_n = (p.N + BN - 1) / BN;
  const int64_t wg = blockIdx.x;
  gemm256_body<T, F3R_A_PLAIN, F3R_EPI_GENERIC, false, 1, F3R_ADD_NONE, 2, LAB>(p, smem, (wg / n_tiles_n) * BM, (int)(wg % n_tiles_n) * BN);
}
template <class T, int LAB>
int launch_lab(const f3r_gemm_args& a, hipStream_t stream) {
  auto kern = gemm256_lab_kernel<T, LAB>;
  constexpr int BN = 256, LDS_BYTES = TileCfg<2>::LDS_BYTES;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  const int64_t tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(NT), LDS_BYTES, stream, a);
  return f3r_check_launch("f3r_gemm(256 lab)");
}
#endif

}  // namespace
