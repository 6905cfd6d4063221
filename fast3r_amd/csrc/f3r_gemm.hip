// f3r_gemm: out = epilogue(A(M,K) * W(N,K)^T) for gfx950.
//
// Tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves (2 x 2), each wave a 64 x 64 sub-tile as 4 x 4
// v_mfma_f32_16x16x32 fragments, fp32 accumulate.  Both operands are K-contiguous ("B^T input"), so
// activations and weights use the same LDS image: [128 rows][64 k] 16-bit, 128 B per row, the 16-byte
// chunk index XOR-swizzled with (row >> 1) & 7 so that every ds_read_b128 lane group (16 lanes: 8
// rows x 2 k-chunks of two row parities) covers all 64 banks once (MI355X_MICROARCH.md, LDS table).
// Global -> LDS goes through registers (16 B per lane, a 128 B row per 8 lanes) because the conv
// gather needs per-lane predication to zero; the loads of k-tile t+1 are issued before the MFMAs of
// tile t and written to the other LDS buffer after them: one barrier per k-tile.
//
// MFMA operand roles: by default the WEIGHT tile is the MFMA "A" operand and the activation tile the
// "B" operand, i.e. D[row = n][col = m]: a lane then owns 4 consecutive n of one output row m, which
// makes every epilogue access (bias, residual, fp32/lowp stores) a 16 B / 8 B vector along n.  The V
// third of the QKV epilogue swaps the roles so that a lane owns 4 consecutive tokens of one channel
// and can write V transposed (vt[d][token]) with 8 B stores.
#include <stdlib.h>

#include "f3r_common.h"
#include "f3r_gemm_epi.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NT = 256;
constexpr int TILE_ELEMS = 128 * 64;                       // one operand tile
constexpr int GEMM_LDS_BYTES = 2 * 2 * TILE_ELEMS * 2;     // 2 stages x (A, W) x 16 KB = 64 KB

__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

struct ConvRow {  // per staged row: output pixel decomposition
  int b, oy, ox;
  bool ok;
};

// GLDS: stage both operand tiles with global_load_lds (16 B per lane straight into LDS, no staging VGPRs, no ds_write
// pass).  The DMA writes wave-uniform-base + lane*16, i.e. a linear image, so the XOR swizzle is applied to the per-lane
// SOURCE address instead (lane l of a wave-instruction fills physical chunk l%8 of row l/8 with logical chunk
// (l%8) ^ ((row>>1)&7)).  Only for the plain operand with K == Kpad (no zero fill needed); rows past M / N are clamped to
// the last valid row (their products are discarded by the epilogue guards).
template <class T, int A_MODE, int EPI, int GLDS>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(const f3r_gemm_args p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* As = smem;                    // [2][128][64]
  uint16_t* Ws = smem + 2 * TILE_ELEMS;   // [2][128][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the per-wave operand roles below stay wave-uniform branches
  const int wm = wid >> 1, wn = wid & 1;

  // XCD-aware tile order: consecutive workgroup ids go round-robin over the 8 XCDs; give each XCD a
  // contiguous run of tiles (same W columns, neighbouring A rows) so its private L2 sees the reuse.
  const int n_tiles_n = (p.N + BN - 1) / BN;
  const int64_t n_tiles_m = (p.M + BM - 1) / BM;
  const int64_t n_wg = n_tiles_m * n_tiles_n;
  int64_t wg = blockIdx.x;
  {
    const int64_t q = n_wg / 8, r = n_wg % 8;
    const int64_t xcd = wg % 8, idx = wg / 8;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // grouped order inside the run: GM m-tiles x all n-tiles per group, m fastest.  The ~64 tiles an XCD runs
  // at once then form an ~8 x 8 patch: 8 A panels + 8 W panels are shared through its 4 MiB L2 instead of
  // every tile streaming its own A panel (A-traffic / n_tiles_n -> / 8 re-reads, served by the Infinity Cache).
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

  // ---- staging roles: 16 B chunk c of rows r0 + 32 i
  const int sc = tid & 7;
  const int sr0 = tid >> 3;
  const uint16_t* Ag = (const uint16_t*)p.A;
  const uint16_t* Wg = (const uint16_t*)p.W;

  ConvRow cr[4];
  int64_t a_row_off[4];
  bool a_row_ok[4];
  bool w_row_ok[4];
  int64_t w_row_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + sr0 + 32 * i;
    a_row_ok[i] = m < p.M;
    if (A_MODE == F3R_A_PLAIN) {
      a_row_off[i] = m * p.lda;
    } else {
      const int64_t per_img = (int64_t)p.conv_OH * p.conv_OW;
      const int64_t mm = a_row_ok[i] ? m : 0;
      cr[i].b = (int)(mm / per_img);
      const int rem = (int)(mm % per_img);
      cr[i].oy = rem / p.conv_OW;
      cr[i].ox = rem % p.conv_OW;
      cr[i].ok = a_row_ok[i];
      a_row_off[i] = 0;
    }
    const int n = n0 + sr0 + 32 * i;
    w_row_ok[i] = n < p.N;
    w_row_off[i] = (int64_t)n * p.Kpad;
  }

  // split precision (f3r.h f3r_split): the K loop runs over nseg segments of nk1 tiles; segment 1 multiplies by the low plane of W,
  // segment 2 takes the low plane of A
  const int nseg = p.split == F3R_SPLIT_NONE ? 1 : (p.split == F3R_SPLIT_W2 ? 2 : 3);
  const int Kpad1 = p.split == F3R_SPLIT_NONE ? p.Kpad : p.Kpad / 2;
  const int nk1 = Kpad1 / BK;
  const int nk = nseg * nk1;
  const int cpad_tiles = (A_MODE == F3R_A_CONV3X3) ? ((p.conv_C + 63) / 64) : 1;
  const uint16_t* Alo = (const uint16_t*)p.A_lo;

  u32x4 ra[4], rw[4];
  auto load_tile = [&](int kt_all) {
    const int seg = kt_all / nk1;
    const int kt = kt_all - seg * nk1;
    const uint16_t* Ag = (seg == 2) ? Alo : (const uint16_t*)p.A;
    if (A_MODE == F3R_A_PLAIN) {
      const int k = kt * BK + sc * 8;
      const bool kok = (k + 8) <= p.K;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (a_row_ok[i] && kok) v = *(const u32x4*)(Ag + a_row_off[i] + k);
        ra[i] = v;
      }
    } else {
      const int tap = kt / cpad_tiles;
      const int ci = (kt - tap * cpad_tiles) * 64 + sc * 8;
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const bool cok = ci < p.conv_C;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = cr[i].oy * p.conv_stride + dy;
        const int ix = cr[i].ox * p.conv_stride + dx;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (cr[i].ok && cok && iy >= 0 && iy < p.conv_H && ix >= 0 && ix < p.conv_W) {
          const int64_t off = (((int64_t)cr[i].b * p.conv_H + iy) * p.conv_W + ix) * p.conv_C + ci;
          v = *(const u32x4*)(Ag + off);
          if (p.a_relu) {
            v[0] = relu_pk(v[0]);
            v[1] = relu_pk(v[1]);
            v[2] = relu_pk(v[2]);
            v[3] = relu_pk(v[3]);
          }
        }
        ra[i] = v;
      }
    }
    const int kw = (seg == 1 ? Kpad1 : 0) + kt * BK + sc * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (w_row_ok[i]) v = *(const u32x4*)(Wg + w_row_off[i] + kw);
      rw[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
    uint16_t* a = As + buf * TILE_ELEMS;
    uint16_t* w = Ws + buf * TILE_ELEMS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = sr0 + 32 * i;
      *(u32x4*)(a + swz(r, sc)) = ra[i];
      *(u32x4*)(w + swz(r, sc)) = rw[i];
    }
  };

  float4v acc[4][4];  // [nf][mf] (default roles) or [mf][nf] (swapped roles)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

  // swapped operand roles only for the V part of the QKV epilogue (wave-uniform: a wave's 64 columns are one head)
  const int Dq_ = p.qkv_dq ? p.qkv_dq : p.N / 3;
  const bool swap_roles = (EPI == F3R_EPI_QKV) && (n0 + wn * 64 >= Dq_ + (p.N - Dq_) / 2);

  const int fr = lane & 15;  // fragment row inside a 16-row block
  const int fg = lane >> 4;  // k-group: 8 elements at k = ks*32 + fg*8

  // operand roles (wave-uniform): opA feeds the MFMA "A" operand (-> D rows), opB the "B" operand (-> D cols)
  const int offA = (swap_roles ? 0 : 2 * TILE_ELEMS);  // default: weights are the A operand
  const int offB = (swap_roles ? 2 * TILE_ELEMS : 0);
  const int rowA = (swap_roles ? wm : wn) * 64 + fr;
  const int rowB = (swap_roles ? wn : wm) * 64 + fr;

  auto compute = [&](int buf) {
    const uint16_t* ta = smem + offA + buf * TILE_ELEMS;
    const uint16_t* tb = smem + offB + buf * TILE_ELEMS;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typename T::vec8 fa[4], fb[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        fa[f] = as_vec8<T>(*(const u32x4*)(ta + swz(rowA + f * 16, ks * 4 + fg)));
        fb[f] = as_vec8<T>(*(const u32x4*)(tb + swz(rowB + f * 16, ks * 4 + fg)));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = T::mfma16(fa[i], fb[j], acc[i][j]);  // acc[A-frag][B-frag]
    }
  };

  if constexpr (GLDS == 1) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    // wave w issues instructions i = 0..3 for each operand: rows (w*4 + i)*8 + lane/8 of the 128-row tile
    const int lrow = lane >> 3, pch = lane & 7;
    const uint16_t* asrc[4];
    const uint16_t* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (wid * 4 + i) * 8 + lrow;
      const int lch = pch ^ ((r >> 1) & 7);
      int64_t m = m0 + r;
      if (m >= p.M) m = p.M - 1;
      int n = n0 + r;
      if (n >= p.N) n = p.N - 1;
      asrc[i] = Ag + m * p.lda + lch * 8;
      wsrc[i] = Wg + (int64_t)n * p.Kpad + lch * 8;
    }
    const int64_t alo_delta = Alo ? (Alo - Ag) : 0;
    auto dma_tile = [&](int kt_all, int buf) {
      const int seg = kt_all / nk1;
      const int kt = kt_all - seg * nk1;
      const int64_t ka = (seg == 2 ? alo_delta : 0) + (int64_t)kt * BK;
      const int kw = (seg == 1 ? Kpad1 : 0) + kt * BK;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint16_t* la = As + buf * TILE_ELEMS + (wid * 4 + i) * 8 * 64;  // wave-uniform: 8 rows x 64 elements per instruction
        uint16_t* lw = Ws + buf * TILE_ELEMS + (wid * 4 + i) * 8 * 64;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(asrc[i] + ka), (lds_ptr_t)la, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(wsrc[i] + kw), (lds_ptr_t)lw, 16, 0, 0);
      }
    };
    dma_tile(0, 0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) dma_tile(kt + 1, cur ^ 1);
      compute(cur);
      __syncthreads();
      cur ^= 1;
    }
  } else {
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = (kt + 1) < nk;
      if (more) load_tile(kt + 1);
      compute(cur);
      if (more) store_tile(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  // ------------------------------------------------------------------ epilogues (f3r_gemm_epi.h)
  const int64_t m_base = m0 + wm * 64;
  const int n_base = n0 + wn * 64;
  if (swap_roles) gemm_epilogue_vt<T, GemmFragLayout<4, 4, 4, 4>, true>(p, &acc[0][0], m_base, n_base, lane);
  else gemm_epilogue_default<T, EPI, GemmFragLayout<4, 4, 4, 4>, true>(p, &acc[0][0], m_base, n_base, lane);
}

template <class T, int A_MODE, int EPI, int GLDS>
int launch(const f3r_gemm_args& a, hipStream_t stream) {
  static bool attr_set = false;  // benign race: idempotent
  auto kern = gemm_kernel<T, A_MODE, EPI, GLDS>;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
    attr_set = true;
  }
  const int64_t tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  if (tiles <= 0) return F3R_OK;
  F3R_REQUIRE(tiles < (1ll << 31), "f3r_gemm: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(NT), GEMM_LDS_BYTES, stream, a);
  return f3r_check_launch("f3r_gemm");
}

template <class T>
int dispatch(const f3r_gemm_args& a, hipStream_t stream) {
  if (a.a_mode == F3R_A_CONV3X3) return launch<T, F3R_A_CONV3X3, F3R_EPI_GENERIC, 0>(a, stream);
  // LDS-DMA staging cannot zero-fill a K tail: plain operands with K < Kpad go through registers
  const int Kpad1 = a.split ? a.Kpad / 2 : a.Kpad;
  const bool glds = a.K == Kpad1 && a.M > 0;
  switch (a.epi) {
    case F3R_EPI_GENERIC: return glds ? launch<T, F3R_A_PLAIN, F3R_EPI_GENERIC, 1>(a, stream) : launch<T, F3R_A_PLAIN, F3R_EPI_GENERIC, 0>(a, stream);
    case F3R_EPI_QKV: return glds ? launch<T, F3R_A_PLAIN, F3R_EPI_QKV, 1>(a, stream) : launch<T, F3R_A_PLAIN, F3R_EPI_QKV, 0>(a, stream);
    case F3R_EPI_CONVT: return launch<T, F3R_A_PLAIN, F3R_EPI_CONVT, 0>(a, stream);
  }
  f3r_set_error("f3r_gemm: bad epi %d", a.epi);
  return F3R_ERR_ARG;
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
inline bool al8(const void* p) { return (((uintptr_t)p) & 7) == 0; }

}  // namespace

extern "C" int f3r_gemm(const f3r_gemm_args* args, f3r_stream_t stream) {
  F3R_REQUIRE(args != nullptr, "f3r_gemm: null args");
  const f3r_gemm_args& a = *args;
  F3R_REQUIRE(a.A && a.W, "f3r_gemm: null operand");
  F3R_REQUIRE(a.M >= 0 && a.N > 0, "f3r_gemm: bad M/N (%lld, %d)", (long long)a.M, a.N);
  F3R_REQUIRE(a.split >= F3R_SPLIT_NONE && a.split <= F3R_SPLIT_X3F8, "f3r_gemm: bad split %d", a.split);
  if (a.split == F3R_SPLIT_W2F8) {  // rows of [K fp16 | K fp8] on both operands: one kernel family takes them (f3r_gemm_asm.hip), nothing else does
    F3R_REQUIRE(al16(a.A) && al16(a.W) && a.w_scale && (((uintptr_t)a.w_scale) & 3) == 0, "f3r_gemm: W2F8 needs 16-byte aligned A / W and w_scale");
    F3R_REQUIRE(a.K > 0 && a.K % 128 == 0 && a.Kpad == a.K, "f3r_gemm: W2F8 needs K = Kpad, a multiple of 128 (got %d / %d)", a.K, a.Kpad);
    F3R_REQUIRE(a.lda % 8 == 0 && a.lda * 2 >= (int64_t)a.K * 3, "f3r_gemm: W2F8 rows are [K fp16 | K fp8]: lda %lld must be >= 3 K / 2", (long long)a.lda);
    F3R_REQUIRE(a.epi != F3R_EPI_GENERIC || a.out_f32 || a.out_lp, "f3r_gemm: no output");
    F3R_REQUIRE(!a.out_f8 && !a.out_relu_f8 && !a.fin_w, "f3r_gemm: W2F8 launches write one output (no out_f8 / out_relu_f8 / fin_w)");
    // the checks the other splits get further down (a C caller reaches the hand-scheduled epilogues only through here: ADVICE r5)
    F3R_REQUIRE((a.kernel_sel >= 0 && a.kernel_sel <= 7) || a.kernel_sel == 9, "f3r_gemm: bad kernel_sel %d", a.kernel_sel);
    F3R_REQUIRE(a.dtype == F3R_F16, "f3r_gemm: W2F8 corrects fp16 planes only (dtype %d)", a.dtype);
    F3R_REQUIRE(a.act >= F3R_ACT_NONE && a.act <= F3R_ACT_RELU, "f3r_gemm: bad act %d", a.act);
    F3R_REQUIRE(a.N % 4 == 0 && (!a.bias || al16(a.bias)), "f3r_gemm: N %d must be a multiple of 4, bias 16-byte aligned", a.N);
    if (a.epi == F3R_EPI_GENERIC) {
      F3R_REQUIRE(!a.out_f32 || (al16(a.out_f32) && a.ldo_f32 % 4 == 0 && a.ldo_f32 >= a.N), "f3r_gemm: out_f32 alignment/ld");
      F3R_REQUIRE(!a.out_lp || (al16(a.out_lp) && a.ldo_lp % 8 == 0 && a.ldo_lp >= a.N), "f3r_gemm: out_lp alignment/ld");
      F3R_REQUIRE(!a.res_f32 || (al16(a.res_f32) && a.ldr_f32 % 4 == 0 && a.ldr_f32 >= a.N), "f3r_gemm: res_f32 alignment/ld");
    }
    const char* why = "";
    if (a.epi == F3R_EPI_QKV) {
      F3R_REQUIRE(a.qkv_dq != 0 || a.N % 3 == 0, "f3r_gemm: QKV with three equal parts needs N %% 3 == 0 (got %d)", a.N);
      {
        const int Dq = a.qkv_dq ? a.qkv_dq : a.N / 3;
        F3R_REQUIRE(Dq > 0 && Dq % 64 == 0 && a.N > Dq && (a.N - Dq) % 128 == 0, "f3r_gemm: QKV parts must be whole 64-wide heads (N %d, q %d)", a.N, Dq);
      }  // q | k with the low plane in fp8, V^T (swapped operand roles) on two fp16 planes from W_aux
      F3R_REQUIRE(a.q && a.k && a.vt && a.W_aux && al16(a.W_aux) && a.seq_len > 0 && a.M % a.seq_len == 0 && a.ldvt >= a.seq_len && a.act == F3R_ACT_NONE,
                  "f3r_gemm: W2F8 QKV needs q, k, vt, W_aux (16-byte aligned), seq_len dividing M, no activation");
      if (a.rope_cos) F3R_REQUIRE(a.rope_sin && a.rope_w > 0 && al16(a.rope_cos) && al16(a.rope_sin), "f3r_gemm: RoPE tables");
      if (!f3r_gemm_asm_qkv_eligible(a, &why)) {
        f3r_set_error("f3r_gemm: split W2F8 QKV but the launch is not eligible for the hand-scheduled kernel: %s", why);
        return F3R_ERR_UNSUPPORTED;
      }
      return f3r_gemm_asm_qkv_launch(a, (hipStream_t)stream);
    }
    if (!f3r_gemm_asm_f8_eligible(a, &why)) {
      f3r_set_error("f3r_gemm: split W2F8 (fp8 low plane) but the launch is not eligible for the hand-scheduled kernel: %s", why);
      return F3R_ERR_UNSUPPORTED;
    }
    return f3r_gemm_asm_f8_launch(a, (hipStream_t)stream);
  }
  // the fp8 output planes and the fused DPT tail belong to the GENERIC epilogue (every other epilogue would silently ignore them)
  F3R_REQUIRE(a.epi == F3R_EPI_GENERIC || (!a.out_f8 && !a.out_relu_f8 && !a.fin_w), "f3r_gemm: out_f8 / out_relu_f8 / fin_w need the generic epilogue (epi %d)", a.epi);
  const int planes = a.split ? 2 : 1;
  F3R_REQUIRE(a.Kpad > 0 && a.Kpad % (64 * planes) == 0, "f3r_gemm: Kpad %d must be a positive multiple of %d", a.Kpad, 64 * planes);
  const int Kpad1 = a.Kpad / planes;
  F3R_REQUIRE((a.split != F3R_SPLIT_X3 && a.split != F3R_SPLIT_X3F8) || (a.A_lo && al16(a.A_lo) && !a.a_relu), "f3r_gemm: X3 / X3F8 split needs A_lo (16-byte aligned) and no a_relu");
  if (a.split == F3R_SPLIT_X3F8)  // fp16 high planes + fp8 correction planes: the 3x3 convolutions of the DPT head on the 256-tile kernel only
    F3R_REQUIRE(a.a_mode == F3R_A_CONV3X3 && a.dtype == F3R_F16 && a.conv_C % 128 == 0 && a.w_scale && (((uintptr_t)a.w_scale) & 3) == 0,
                "f3r_gemm: X3F8 needs a 3x3 convolution with conv_C %% 128 == 0 (got %d), fp16 planes and w_scale", a.conv_C);
  F3R_REQUIRE((a.kernel_sel >= 0 && a.kernel_sel <= 7) || a.kernel_sel == 9 || a.kernel_sel >= 16, "f3r_gemm: bad kernel_sel %d", a.kernel_sel);
  F3R_REQUIRE(a.N % 4 == 0, "f3r_gemm: N %d must be a multiple of 4", a.N);
  F3R_REQUIRE(al16(a.A) && al16(a.W), "f3r_gemm: A/W must be 16-byte aligned");
  F3R_REQUIRE(a.dtype == F3R_F16 || a.dtype == F3R_BF16, "f3r_gemm: bad dtype %d", a.dtype);
  if (a.a_mode == F3R_A_PLAIN) {
    F3R_REQUIRE(a.K > 0 && a.K % 8 == 0 && a.K <= Kpad1, "f3r_gemm: K %d must be a multiple of 8 and <= Kpad", a.K);
    F3R_REQUIRE(a.lda % 8 == 0 && a.lda >= a.K, "f3r_gemm: lda %lld must be a multiple of 8 and >= K", (long long)a.lda);
  } else if (a.a_mode == F3R_A_CONV3X3) {
    F3R_REQUIRE(a.conv_C > 0 && a.conv_C % 8 == 0, "f3r_gemm: conv_C %d must be a multiple of 8", a.conv_C);
    F3R_REQUIRE(Kpad1 == 9 * ((a.conv_C + 63) / 64) * 64, "f3r_gemm: conv Kpad %d != 9*roundup(C,64) per plane", a.Kpad);
    F3R_REQUIRE(a.epi == F3R_EPI_GENERIC, "f3r_gemm: conv3x3 supports only the generic epilogue");
    F3R_REQUIRE(a.conv_stride == 1 || a.conv_stride == 2, "f3r_gemm: conv stride %d", a.conv_stride);
    F3R_REQUIRE(a.conv_OH == (a.conv_H + 2 - 3) / a.conv_stride + 1 && a.conv_OW == (a.conv_W + 2 - 3) / a.conv_stride + 1,
                "f3r_gemm: conv output dims inconsistent");
    F3R_REQUIRE(a.conv_OH > 0 && a.conv_OW > 0 && a.M % ((int64_t)a.conv_OH * a.conv_OW) == 0, "f3r_gemm: conv M not a multiple of OH*OW");
  } else {
    f3r_set_error("f3r_gemm: bad a_mode %d", a.a_mode);
    return F3R_ERR_ARG;
  }
  if (a.epi == F3R_EPI_GENERIC) {
    F3R_REQUIRE(a.out_f32 || a.out_lp || a.fin_w, "f3r_gemm: no output");
    F3R_REQUIRE(!a.out_f32 || (al16(a.out_f32) && a.ldo_f32 % 4 == 0 && a.ldo_f32 >= a.N), "f3r_gemm: out_f32 alignment/ld");
    F3R_REQUIRE(!a.out_lp || (al8(a.out_lp) && a.ldo_lp % 4 == 0 && a.ldo_lp >= a.N), "f3r_gemm: out_lp alignment/ld");
    F3R_REQUIRE((!a.out_f8 && !a.out_relu_f8) || (a.dtype == F3R_F16 && a.N % 8 == 0 && al8(a.out_f8) && al8(a.out_relu_f8) && !a.out_lp_f8),
                "f3r_gemm: out_f8 / out_relu_f8 need fp16, N %% 8 == 0 and 8-byte aligned buffers");
    if (a.fin_w) {
      F3R_REQUIRE(a.fin_b && a.fin_pts && al16(a.fin_w) && (a.fin_n_out == 3 || a.fin_n_out == 4) && (a.fin_n_out == 4 || !a.fin_conf),
                  "f3r_gemm: fin_w needs fin_b, fin_pts, a 16-byte aligned [4][N] weight and fin_n_out 3 or 4 (a confidence output needs 4)");
      F3R_REQUIRE(a.fin_depth_mode >= 0 && a.fin_depth_mode <= 2 && (a.fin_conf_mode == 0 || a.fin_conf_mode == 1), "f3r_gemm: bad fin_depth_mode / fin_conf_mode");
      F3R_REQUIRE(!a.out_f32 && !a.out_relu && !a.out_f8 && !a.out_relu_f8 && !a.res_lp && !a.res_f32 && !a.rowadd,
                  "f3r_gemm: the fused head tail writes fin_pts / fin_conf only");
    }
    F3R_REQUIRE(!a.res_f32 || (al16(a.res_f32) && a.ldr_f32 % 4 == 0), "f3r_gemm: res_f32 alignment/ld");
    F3R_REQUIRE(!a.res_lp || (al8(a.res_lp) && a.ldr_lp % 4 == 0), "f3r_gemm: res_lp alignment/ld");
    F3R_REQUIRE(!a.res_lp2 || (al8(a.res_lp2) && a.ldr_lp2 % 4 == 0), "f3r_gemm: res_lp2 alignment/ld");
    F3R_REQUIRE((!a.res_lp_lo || (a.res_lp && al8(a.res_lp_lo))) && (!a.res_lp2_lo || (a.res_lp2 && al8(a.res_lp2_lo))), "f3r_gemm: low residual planes need their high planes");
    F3R_REQUIRE((!a.out_lp_lo || (a.out_lp && al8(a.out_lp_lo))) && (!a.out_relu || (al8(a.out_relu) && a.ldo_lp % 4 == 0 && a.ldo_lp >= a.N)) &&
                (!a.out_relu_lo || (a.out_relu && al8(a.out_relu_lo))), "f3r_gemm: out_lp_lo / out_relu alignment");
    F3R_REQUIRE(!a.rowadd || (al16(a.rowadd) && a.rowadd_div > 0), "f3r_gemm: rowadd alignment/div");
    const int64_t ld_max = 1ll << 26;  // the epilogues address a row as 32-bit byte offsets: 15 rows * ld * 4 B < 2^32
    F3R_REQUIRE((!a.out_f32 || a.ldo_f32 < ld_max) && (!a.out_lp || a.ldo_lp < ld_max) && (!a.res_f32 || (a.ldr_f32 >= 0 && a.ldr_f32 < ld_max)) &&
                (!a.res_lp || (a.ldr_lp >= 0 && a.ldr_lp < ld_max)) && (!a.res_lp2 || (a.ldr_lp2 >= 0 && a.ldr_lp2 < ld_max)), "f3r_gemm: row strides must be < 2^26 elements");
    F3R_REQUIRE(!a.bias || al16(a.bias), "f3r_gemm: bias alignment");
  } else if (a.epi == F3R_EPI_QKV) {
    const int Dq = a.qkv_dq ? a.qkv_dq : a.N / 3;
    F3R_REQUIRE(a.qkv_dq != 0 || a.N % 3 == 0, "f3r_gemm: QKV with three equal parts needs N %% 3 == 0 (got %d)", a.N);
    F3R_REQUIRE(Dq > 0 && Dq % 64 == 0 && a.N > Dq && (a.N - Dq) % 128 == 0, "f3r_gemm: QKV parts must be whole 64-wide heads (N %d, q %d)", a.N, Dq);
    F3R_REQUIRE(a.q && a.k && a.vt && al8(a.q) && al8(a.k) && al8(a.vt), "f3r_gemm: QKV outputs null/misaligned");
    F3R_REQUIRE(a.seq_len > 0 && a.M % a.seq_len == 0 && a.ldvt >= a.seq_len, "f3r_gemm: QKV seq_len/ldvt");
    F3R_REQUIRE(a.act == F3R_ACT_NONE, "f3r_gemm: QKV has no activation");
    F3R_REQUIRE(!a.bias || al16(a.bias), "f3r_gemm: bias alignment");
    if (a.rope_cos) F3R_REQUIRE(a.rope_sin && a.rope_w > 0 && al16(a.rope_cos) && al16(a.rope_sin), "f3r_gemm: RoPE tables");
  } else if (a.epi == F3R_EPI_CONVT) {
    F3R_REQUIRE(a.out_lp && al8(a.out_lp), "f3r_gemm: CONVT needs out_lp");
    F3R_REQUIRE(a.ct_s > 0 && a.ct_cout > 0 && a.ct_cout % 4 == 0 && a.N == a.ct_s * a.ct_s * a.ct_cout, "f3r_gemm: CONVT N != s*s*cout");
    F3R_REQUIRE(a.ct_h > 0 && a.ct_w > 0 && a.M % ((int64_t)a.ct_h * a.ct_w) == 0, "f3r_gemm: CONVT M not a multiple of h*w");
    F3R_REQUIRE(!a.bias || al16(a.bias), "f3r_gemm: bias alignment");
  }
  hipStream_t s = (hipStream_t)stream;
  // kernel_sel: 0 = by shape (256-tile kernel for the large, regular problems), 1 = 128-tile kernel, 2 / 3 = 256-tile kernel with /
  // without the staggered wave rows (measurement; an ineligible shape is an error, not a silent fallback)
  if (a.kernel_sel >= 16) return f3r_gemm256_lab(a, s);
  if (a.split == F3R_SPLIT_X3F8 || a.fin_w) {  // one kernel family takes these (f3r_gemm256_f8.hip): no second path, no fallback
    if (!f3r_gemm256_eligible(a)) {
      f3r_set_error("f3r_gemm: split X3F8 / fin_w but the launch is not eligible for the 256-tile kernel (conv_C %% 64 (X3F8: 128) == 0, "
                    "N %% 128 == 0 (fin_w: N == 128), operand below 4 GiB)");
      return F3R_ERR_UNSUPPORTED;
    }
    return f3r_gemm256_launch(a, s, 1);
  }
  // 6 = the hand-scheduled one-wave-per-SIMD kernel (csrc/asm/gemm_gen.py), an ineligible launch is an error; 0 takes it for the launches
  // it is built for (the transformer's big linear layers: enough full 256 x 256 tiles to fill its persistent grid, f3r_gemm_asm_preferred);
  // 7 = automatic WITHOUT it
  if (a.kernel_sel == 6 || a.kernel_sel == 0 || a.kernel_sel == 9) {
    const char* why = "";
    // (QKV = two launches: q | k with 2/3 of the tiles, V^T with 1/3; the smaller one decides)
    if (a.epi == F3R_EPI_QKV && f3r_gemm_asm_qkv_eligible(a, &why) && (a.kernel_sel == 6 || a.kernel_sel == 9 || f3r_gemm_asm_preferred((a.M / 256) * (a.N / 3 / 256))))
      return f3r_gemm_asm_qkv_launch(a, s);
    const bool ok = a.epi != F3R_EPI_QKV && f3r_gemm_asm_eligible(a, &why);
    if ((a.kernel_sel == 6 || a.kernel_sel == 9) && !ok) {
      f3r_set_error("f3r_gemm: kernel_sel 6 (hand-scheduled kernel) but the launch is not eligible: %s", why);
      return F3R_ERR_UNSUPPORTED;
    }
    if (ok && (a.kernel_sel == 6 || a.kernel_sel == 9 || f3r_gemm_asm_preferred((a.M / 256) * (a.N / 256)))) return f3r_gemm_asm_launch(a, s);
  }
  const int sel = a.kernel_sel == 7 ? 0 : a.kernel_sel;
  if (sel >= 2) F3R_REQUIRE(f3r_gemm256_eligible(a), "f3r_gemm: kernel_sel %d but the shape is not eligible for the 256-tile kernel", sel);
  if (sel >= 2 || (sel == 0 && f3r_gemm256_eligible(a) && f3r_gemm256_preferred(a))) return f3r_gemm256_launch(a, s, sel != 3);
  return a.dtype == F3R_F16 ? dispatch<F16>(a, s) : dispatch<BF16>(a, s);
}
