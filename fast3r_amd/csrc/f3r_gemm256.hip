// f3r_gemm256: the large-shape path of f3r_gemm -- out = epilogue(A(M,K) * W(N,K)^T) on a 256 x 256 x 64 tile for gfx950.
//
// One 512-thread workgroup (8 waves, 2 per SIMD) per CU, 128 KiB of LDS: two K-tile buffers of four 16 KiB HALF TILES each
// (A rows 0-127 / 128-255, W rows 0-127 / 128-255; [128 rows][64 k] 16-bit, 128-byte rows, 16-byte chunk c of row r stored at chunk
// c ^ ((r >> 1) & 7): every ds_read_b128 lane group covers the 64 banks exactly once).  Waves are 2 (M) x 4 (N); a wave owns a
// 128 x 64 output sub-tile = 8 x 4 fragments of v_mfma_f32_16x16x32 (128 fp32 accumulator registers) and therefore reads ONE A half
// tile (wm) and one W half tile (wn >> 1).
//
// Schedule (cdna_hip_programming.md "256^2 8-phase", re-derived here because every wait below is placed by counting):
//   * a K-tile is 4 PHASES, one 64 x 32 output quadrant x K = 64 each (16 MFMAs); fragment reads 12 / 8 / 4 / 0 ds_read_b128 per
//     phase = 24 per 64 MFMAs (0.375 per MFMA): quadrants (A0,W0) (A1,W0) (A1,W1) (A0,W1) keep both A halves in registers;
//   * all global -> LDS traffic is LDS-DMA (global_load_lds, 16 B per lane, no staging registers): ONE half tile (2 instructions per
//     wave) per phase, issued at least a K-tile ahead for the streamed operand:
//         phase 0 of tile t:  A half 1 of tile t+1        phase 1:  W half 0 of tile t+1
//         phase 2          :  W half 1 of tile t+1        phase 3:  A half 0 of tile t+2, then s_waitcnt vmcnt(2)
//     so the DMA queue is never drained in the loop (vmcnt(2) leaves the half tile just issued in flight across the barriers) and an
//     A half tile has ~4 phases (~2000 cycles) to arrive -- HBM latency -- while the L2-resident W has ~2;
//   * the two wave rows (wm = 0 / 1, one wave of each per SIMD) run STAGGERED by one barrier: while one does its 16 MFMAs (s_setprio 1)
//     the other issues its ds_reads and LDS-DMA, so the matrix pipe and the LDS / TA pipes alternate owners instead of colliding;
//   * every phase is  [ds_reads, LDS-DMA, (vmcnt)] s_barrier [lgkmcnt(0), 16 MFMA] s_barrier.
// Hazards, with the stagger (a wave of row 1 is one barrier behind a wave of row 0):
//   RAW  LDS-DMA data may be read one phase after the phase whose FIRST barrier follows the issuers' vmcnt wait: the wait sits in
//        phase 3 before its first barrier, the first read of the tile in phase 0 of the next tile;
//   WAR  a half tile may be re-staged two phases after the phase that issued its last read: A halves (last read: phase 1) from
//        phase 3, W halves (last read: phase 2) from phase 0 of the next tile -- the table above restages A at phase 3 / 0 and W at 1 / 2.
//
// Operand roles, epilogues, the split-precision K segments and the LDS swizzle are those of f3r_gemm.hip; the implicit-GEMM 3x3
// convolution stages its operand by LDS-DMA too: out-of-image taps read a 16-byte zero line instead of being predicated.
#include "f3r_common.h"
#include "f3r_gemm_epi.h"

__device__ __attribute__((aligned(128))) uint32_t f3r_zero_line[32];  // 128 B of zeros: the source of every padded conv tap

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NT = 512;
constexpr int HT = 128 * 64;          // elements of a half tile
constexpr int BUF = 4 * HT;           // A_h0 A_h1 W_h0 W_h1
constexpr int LDS_BYTES = 2 * BUF * 2;  // 131072

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int V>
struct IC {
  static constexpr int value = V;
};

#define F3R_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))   /* vmcnt(n), n <= 15; lgkmcnt / expcnt untouched */
#define F3R_LGKMCNT0() __builtin_amdgcn_s_waitcnt(0xC07F)      /* lgkmcnt(0); vmcnt untouched */

template <class T, int A_MODE, int EPI, bool SWAP, int STAGGER, int ADDSRC>
__device__ __forceinline__ void gemm256_body(const f3r_gemm_args& p, uint16_t* smem, int64_t m0, int n0) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int fr = lane & 15, fg = lane >> 4;

  // ------------------------------------------------------------------ K segments (split precision) and tile counts
  const int nseg = p.split == F3R_SPLIT_NONE ? 1 : (p.split == F3R_SPLIT_W2 ? 2 : 3);
  const int Kpad1 = p.split == F3R_SPLIT_NONE ? p.Kpad : p.Kpad / 2;
  const int nk1 = Kpad1 / BK;
  const int nk = nseg * nk1;
  const int ctiles = A_MODE == F3R_A_CONV3X3 ? p.conv_C / 64 : 1;

  // ------------------------------------------------------------------ LDS-DMA source addressing
  // wave w, instruction i of a half tile: rows (w*2 + i)*8 + lane/8, physical chunk lane%8 <- logical chunk (lane%8) ^ ((row>>1)&7)
  uint32_t a_off[2][2], w_off[2][2];  // byte offsets of this lane's 16 B inside the tile's operand panel [half][i]
  uint32_t a_msk[2][2];               // CONV: bit tap = the tap of this lane's pixel lies inside the image
  const char* const Ab = (const char*)p.A;
  const char* const Alo = (const char*)p.A_lo;
  const char* const Wb = (const char*)p.W + (int64_t)n0 * p.Kpad * 2;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wid * 2 + i) * 8 + (lane >> 3);
      const int lc = (lane & 7) ^ ((r >> 1) & 7);
      int n = n0 + h * 128 + r;
      if (n >= p.N) n = p.N - 1;
      w_off[h][i] = (uint32_t)(((int64_t)(n - n0) * p.Kpad + lc * 8) * 2);
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
        a_off[h][i] = (uint32_t)(((((int64_t)b * p.conv_H + oy) * p.conv_W + ox) * p.conv_C + lc * 8) * 2);
        uint32_t msk = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
          if (ok && iy >= 0 && iy < p.conv_H && ix >= 0 && ix < p.conv_W) msk |= 1u << tap;
        }
        a_msk[h][i] = msk;
      }
    }

  // cursors: which K-tile the NEXT A / W half-tile pair is loaded for (wave-uniform; clamped at the last tile, see the loop tail)
  int a_seg = 0, a_kk = 0, a_tap = 0, a_ct = 0, a_t = 0;
  int w_seg = 0, w_kk = 0, w_t = 0;
  auto a_advance = [&]() {
    if (a_t + 1 < nk) {
      ++a_t; ++a_kk; ++a_ct;
      if (a_ct == ctiles) { a_ct = 0; ++a_tap; }
      if (a_kk == nk1) { a_kk = 0; a_tap = 0; a_ct = 0; ++a_seg; }
    }
  };
  auto w_advance = [&]() {
    if (w_t + 1 < nk) {
      ++w_t; ++w_kk;
      if (w_kk == nk1) { w_kk = 0; ++w_seg; }
    }
  };
  auto issue_a = [&](int h, int buf) {  // A half tile h of the cursor's K-tile -> buffer buf
    const char* plane = (a_seg == 2) ? Alo : Ab;
    uint16_t* dst = smem + buf * BUF + h * HT + wid * 2 * 8 * 64;
    if (A_MODE == F3R_A_PLAIN) {
      const char* base = plane + (m0 * p.lda + (int64_t)a_kk * BK) * 2;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + a_off[h][i]), (lds_ptr_t)(dst + i * 8 * 64), 16, 0, 0);
    } else {
      const int dy = a_tap / 3 - 1, dx = a_tap - (a_tap / 3) * 3 - 1;
      const char* base = plane + (((int64_t)dy * p.conv_W + dx) * p.conv_C + a_ct * 64) * 2;
      const char* zl = (const char*)f3r_zero_line;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const char* src = ((a_msk[h][i] >> a_tap) & 1u) ? base + a_off[h][i] : zl;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(dst + i * 8 * 64), 16, 0, 0);
      }
    }
  };
  auto issue_w = [&](int h, int buf) {
    const char* base = Wb + ((int64_t)(w_seg == 1 ? Kpad1 : 0) + (int64_t)w_kk * BK) * 2;
    uint16_t* dst = smem + buf * BUF + (2 + h) * HT + wid * 2 * 8 * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + w_off[h][i]), (lds_ptr_t)(dst + i * 8 * 64), 16, 0, 0);
  };

  // ------------------------------------------------------------------ fragment read addressing (elements inside a buffer)
  const int sw = (fr >> 1) & 7;
  int a_rd[2], w_rd[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int pc = ((ks * 4 + fg) ^ sw) << 3;
    a_rd[ks] = wm * HT + fr * 64 + pc;
    w_rd[ks] = (2 + (wn >> 1)) * HT + ((wn & 1) * 64 + fr) * 64 + pc;
  }

  float4v acc[32];
  typename T::vec8 fa0[2][4], fa1[2][4], fw[2][2];

  auto read_a = [&](typename T::vec8 (&f)[2][4], int buf, int mh) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mf = 0; mf < 4; ++mf)
        f[ks][mf] = as_vec8<T>(*(const u32x4*)(smem + buf * BUF + a_rd[ks] + (mh * 64 + mf * 16) * 64));
  };
  auto read_w = [&](int buf, int nh) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
        fw[ks][nf] = as_vec8<T>(*(const u32x4*)(smem + buf * BUF + w_rd[ks] + (nh * 32 + nf * 16) * 64));
  };
  auto mma = [&](const typename T::vec8 (&f)[2][4], int mh, int nh) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
          const int NF = nh * 2 + nf, MF = mh * 4 + mf;
          if (SWAP) acc[MF * 4 + NF] = T::mfma16(f[ks][mf], fw[ks][nf], acc[MF * 4 + NF]);
          else      acc[NF * 8 + MF] = T::mfma16(fw[ks][nf], f[ks][mf], acc[NF * 8 + MF]);
        }
  };

  // one K-tile out of buffer B (compile-time), 4 phases
  auto tile = [&](auto bufc) {
    constexpr int B = decltype(bufc)::value;
    // ---- phase 0: quadrant (A0, W0)
    read_w(B, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(fa0, B, 0);
    issue_a(1, B ^ 1);
    a_advance();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    F3R_LGKMCNT0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma(fa0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 1: quadrant (A1, W0)
    read_a(fa1, B, 1);
    issue_w(0, B ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    F3R_LGKMCNT0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma(fa1, 1, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: quadrant (A1, W1)
    read_w(B, 1);
    issue_w(1, B ^ 1);
    w_advance();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    F3R_LGKMCNT0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma(fa1, 1, 1);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: quadrant (A0, W1); A half 0 two tiles ahead, then retire everything older (= all of the next tile)
    issue_a(0, B);
    F3R_VMCNT(2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma(fa0, 0, 1);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  // ------------------------------------------------------------------ prologue: all of tile 0 and A half 0 of tile 1
  // The bias and the additive epilogue terms (fp32 / lowp residuals, image-id rows) are loaded FIRST, straight into the accumulators:
  // they land under the latency of the first tiles (f3r_gemm_epi.h); the compiler's own wait covers their first use.
  gemm_acc_init_additive<T, 4, 8, ADDSRC, SWAP>(p, acc, m0 + wm * 128, n0 + wn * 64, lane);
  issue_a(0, 0);
  issue_a(1, 0);
  a_advance();
  issue_w(0, 0);
  issue_w(1, 0);
  w_advance();
  issue_a(0, 1);
  F3R_VMCNT(2);
  __builtin_amdgcn_s_barrier();
  if (STAGGER && wm == 1) __builtin_amdgcn_s_barrier();

  for (int t = 0; t < nk; t += 2) {
    tile(IC<0>{});
    if (t + 1 < nk) tile(IC<1>{});
  }
  // Past the last tile the cursors stay clamped, so the tail re-loads the last tile into half tiles nobody reads any more; drain them
  // before the workgroup's LDS can be handed to the next one.
  F3R_VMCNT(0);
  if (STAGGER && wm == 0) __builtin_amdgcn_s_barrier();

  // ------------------------------------------------------------------ epilogue
  const int64_t m_base = m0 + wm * 128;
  const int n_base = n0 + wn * 64;
  if (SWAP) gemm_epilogue_vt<T, 4, 8, false>(p, acc, m_base, n_base, lane);
  else gemm_epilogue_default<T, EPI, 4, 8, false>(p, acc, m_base, n_base, lane);
}

template <class T, int A_MODE, int EPI, int STAGGER, int ADDSRC>
__global__ __launch_bounds__(NT, 1) void gemm256_kernel(const f3r_gemm_args p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  // XCD-aware tile order: consecutive workgroup ids go round-robin over the 8 XCDs; give each XCD a contiguous run of tiles, and
  // inside the run walk GM m-tiles x all n-tiles with m fastest, so the ~32 tiles an XCD runs at once share 8 A panels and all of W
  // through its private 4 MiB L2.
  const int n_tiles_n = (p.N + BN - 1) / BN;
  const int64_t n_tiles_m = (p.M + BM - 1) / BM;
  const int64_t n_wg = n_tiles_m * n_tiles_n;
  int64_t wg = blockIdx.x;
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
  const int64_t m0 = tm * BM;
  const int n0 = tn * BN;
  if (EPI == F3R_EPI_QKV && n0 >= 2 * (p.N / 3))
    gemm256_body<T, A_MODE, EPI, true, STAGGER, F3R_ADD_NONE>(p, smem, m0, n0);
  else
    gemm256_body<T, A_MODE, EPI, false, STAGGER, ADDSRC>(p, smem, m0, n0);
}

template <class T, int A_MODE, int EPI, int STAGGER, int ADDSRC>
int launch256(const f3r_gemm_args& a, hipStream_t stream) {
  static bool attr_set = false;  // benign race: idempotent
  auto kern = gemm256_kernel<T, A_MODE, EPI, STAGGER, ADDSRC>;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  const int64_t tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(NT), LDS_BYTES, stream, a);
  return f3r_check_launch("f3r_gemm(256)");
}

template <class T>
int dispatch256(const f3r_gemm_args& a, hipStream_t stream, int stagger) {
#define F3R_L256(AM, EP, AD) (stagger ? launch256<T, AM, EP, 1, AD>(a, stream) : launch256<T, AM, EP, 0, AD>(a, stream))
  const int add = gemm_additive_pattern(a);
  if (a.a_mode == F3R_A_CONV3X3) return add == F3R_ADD_RES_LP ? F3R_L256(F3R_A_CONV3X3, F3R_EPI_GENERIC, F3R_ADD_RES_LP) : F3R_L256(F3R_A_CONV3X3, F3R_EPI_GENERIC, F3R_ADD_NONE);
  switch (a.epi) {
    case F3R_EPI_GENERIC:
      return add == F3R_ADD_RES_F32 ? F3R_L256(F3R_A_PLAIN, F3R_EPI_GENERIC, F3R_ADD_RES_F32)
           : add == F3R_ADD_ROWADD ? F3R_L256(F3R_A_PLAIN, F3R_EPI_GENERIC, F3R_ADD_ROWADD) : F3R_L256(F3R_A_PLAIN, F3R_EPI_GENERIC, F3R_ADD_NONE);
    case F3R_EPI_QKV: return F3R_L256(F3R_A_PLAIN, F3R_EPI_QKV, F3R_ADD_NONE);
    default: return F3R_L256(F3R_A_PLAIN, F3R_EPI_CONVT, F3R_ADD_NONE);
  }
#undef F3R_L256
}

}  // namespace

// Whether the 256-tile kernel takes this (already validated) problem: everything its LDS-DMA staging cannot express -- K tails, ragged
// channel counts, strided or pre-activated conv operands, small or narrow outputs -- stays on the 128-tile kernel.
bool f3r_gemm256_eligible(const f3r_gemm_args& a) {
  if (a.M < 2048 || a.N % 128 != 0 || a.N < 256) return false;
  // additive epilogue terms enter through the accumulators: one kind at a time, no activation in between, and only the kinds the
  // model uses on each operand mode (fp32 residual / image-id rows on plain GEMMs, lowp skip connections on convolutions)
  const int add = gemm_additive_pattern(a);
  if (add == F3R_ADD_UNSUPPORTED || (add != F3R_ADD_NONE && a.act != F3R_ACT_NONE)) return false;
  if (add != F3R_ADD_NONE && a.epi != F3R_EPI_GENERIC) return false;
  if (a.a_mode == F3R_A_CONV3X3 ? (add == F3R_ADD_RES_F32 || add == F3R_ADD_ROWADD) : add == F3R_ADD_RES_LP) return false;
  const int Kpad1 = a.split ? a.Kpad / 2 : a.Kpad;
  if (Kpad1 % 64 != 0) return false;
  if (a.a_mode == F3R_A_PLAIN) {
    if (a.K != Kpad1) return false;
    if ((int64_t)256 * a.lda * 2 >= (1ll << 32)) return false;
  } else {
    if (a.conv_stride != 1 || a.a_relu || a.conv_C % 64 != 0) return false;
    if (a.M * (int64_t)a.conv_C * 2 >= (1ll << 32)) return false;  // 32-bit byte offsets into the NHWC operand
  }
  if ((int64_t)256 * a.Kpad * 2 >= (1ll << 32)) return false;
  if (a.epi == F3R_EPI_QKV && (a.N / 3) % 256 != 0) return false;
  return true;
}

int f3r_gemm256_launch(const f3r_gemm_args& a, hipStream_t stream, int stagger) {
  const int64_t tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  if (tiles <= 0) return F3R_OK;
  F3R_REQUIRE(tiles < (1ll << 31), "f3r_gemm: grid too large");
  return a.dtype == F3R_F16 ? dispatch256<F16>(a, stream, stagger) : dispatch256<BF16>(a, stream, stagger);
}
