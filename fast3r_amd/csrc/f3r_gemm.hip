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

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NT = 256;
constexpr int TILE_ELEMS = 128 * 64;                       // one operand tile
constexpr int GEMM_LDS_BYTES = 2 * 2 * TILE_ELEMS * 2;     // 2 stages x (A, W) x 16 KB = 64 KB

__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

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
  const int wid = tid >> 6;
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

  const int nk = p.Kpad / BK;
  const int cpad_tiles = (A_MODE == F3R_A_CONV3X3) ? ((p.conv_C + 63) / 64) : 1;

  u32x4 ra[4], rw[4];
  auto load_tile = [&](int kt) {
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
    const int kw = kt * BK + sc * 8;
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

  // swapped operand roles only for the V third of the QKV epilogue (block-uniform)
  const int Dm = p.N / 3;
  const bool swap_roles = (EPI == F3R_EPI_QKV) && (n0 >= 2 * Dm);

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

  if constexpr (GLDS == 2) {
    // ---- 4-stage ring of 32-deep k-tiles, counted vmcnt: three tiles of DMA stay in flight across the barriers, so the
    // HBM/L2 latency of a tile is covered by the MFMAs of the two tiles in front of it (cdna_hip_programming.md T3/T4).
    // Stage = A [128][32] | W [128][32] = 16 KB; 64-byte rows; 16-byte chunk c of row r lives at chunk c ^ ((-(r >> 2)) & 3)
    // (conflict-free for the ds_read_b128 lane groups at this row stride).  A wave-instruction of the DMA fills 16 rows.
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    constexpr int ST = 4, SE = 128 * 32;  // stages, elements per operand per stage
    uint16_t* const A4 = smem;            // [ST][128][32]
    uint16_t* const W4 = smem + ST * SE;  // [ST][128][32]
    const int lrow = lane >> 2, pch = lane & 3;
    const uint16_t* asrc[2];
    const uint16_t* wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wid * 2 + i) * 16 + lrow;
      const int lch = pch ^ ((-(r >> 2)) & 3);
      int64_t m = m0 + r;
      if (m >= p.M) m = p.M - 1;
      int n = n0 + r;
      if (n >= p.N) n = p.N - 1;
      asrc[i] = Ag + m * p.lda + lch * 8;
      wsrc[i] = Wg + (int64_t)n * p.Kpad + lch * 8;
    }
    auto dma = [&](int kt32) {
      const int st = kt32 & (ST - 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(asrc[i] + kt32 * 32), (lds_ptr_t)(A4 + st * SE + (wid * 2 + i) * 16 * 32), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(wsrc[i] + kt32 * 32), (lds_ptr_t)(W4 + st * SE + (wid * 2 + i) * 16 * 32), 16, 0, 0);
      }
    };
    const int nk32 = p.Kpad / 32;
    const int offA4 = swap_roles ? 0 : ST * SE, offB4 = swap_roles ? ST * SE : 0;
    int foff[4];  // fragment offsets inside a stage: row (f*16 + fr) of the wave's 64-row slice, k-chunk fg
#pragma unroll
    for (int f = 0; f < 4; ++f) foff[f] = (f * 16 + fr) * 32 + ((fg ^ ((-((f * 16 + fr) >> 2)) & 3)) << 3);
    const int rbaseA = (swap_roles ? wm : wn) * 64 * 32, rbaseB = (swap_roles ? wn : wm) * 64 * 32;
#pragma unroll
    for (int t = 0; t < ST - 1; ++t)
      if (t < nk32) dma(t);
    for (int kt = 0; kt < nk32; ++kt) {
      // tile kt must have landed: at most the DMAs of the (ST-2) younger tiles may still be in flight
      if (kt + ST - 2 < nk32) __builtin_amdgcn_s_waitcnt(0x0F70 | 8);   // vmcnt(8) = 2 tiles x 4 DMA instructions per wave
      else if (kt + 1 < nk32) __builtin_amdgcn_s_waitcnt(0x0F70 | 4);   // tail: one younger tile
      else __builtin_amdgcn_s_waitcnt(0x0F70);                          // last tile: everything
      __builtin_amdgcn_s_barrier();  // every wave's share of tile kt has landed AND every wave is done reading tile kt-1
      if (kt + ST - 1 < nk32) dma(kt + ST - 1);  // refill the slot tile kt-1 just left
      const int st = kt & (ST - 1);
      const uint16_t* ta = smem + offA4 + st * SE + rbaseA;
      const uint16_t* tb = smem + offB4 + st * SE + rbaseB;
      typename T::vec8 fa[4], fb[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        fa[f] = as_vec8<T>(*(const u32x4*)(ta + foff[f]));
        fb[f] = as_vec8<T>(*(const u32x4*)(tb + foff[f]));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = T::mfma16(fa[i], fb[j], acc[i][j]);
    }
    __syncthreads();
  } else if constexpr (GLDS == 1) {
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
    auto dma_tile = [&](int kt, int buf) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint16_t* la = As + buf * TILE_ELEMS + (wid * 4 + i) * 8 * 64;  // wave-uniform: 8 rows x 64 elements per instruction
        uint16_t* lw = Ws + buf * TILE_ELEMS + (wid * 4 + i) * 8 * 64;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(asrc[i] + kt * BK), (lds_ptr_t)la, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(wsrc[i] + kt * BK), (lds_ptr_t)lw, 16, 0, 0);
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

  // ------------------------------------------------------------------ epilogues
  if (EPI == F3R_EPI_GENERIC || EPI == F3R_EPI_CONVT) {
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
      const int64_t m = m0 + wm * 64 + mf * 16 + fr;
      if (m >= p.M) continue;
      int64_t ct_base = 0;
      if (EPI == F3R_EPI_CONVT) {
        const int hw = p.ct_h * p.ct_w;
        const int b = (int)(m / hw);
        const int rem = (int)(m % hw);
        const int y = rem / p.ct_w, x = rem % p.ct_w;
        ct_base = (((int64_t)b * p.ct_h * p.ct_s + (int64_t)y * p.ct_s) * ((int64_t)p.ct_w * p.ct_s) + (int64_t)x * p.ct_s);
      }
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        const int nb = n0 + wn * 64 + nf * 16 + fg * 4;
        if (nb >= p.N) continue;
        float4v v = acc[nf][mf];
        if (p.bias) {
          const float4v bb = *(const float4v*)(p.bias + nb);
          v += bb;
        }
        if (p.act == F3R_ACT_GELU) {
          v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]);
        } else if (p.act == F3R_ACT_RELU) {
          v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
        }
        if (EPI == F3R_EPI_GENERIC) {
          if (p.rowadd) v += *(const float4v*)(p.rowadd + (m / p.rowadd_div) * (int64_t)p.N + nb);
          if (p.res_f32) v += *(const float4v*)(p.res_f32 + m * p.ldr_f32 + nb);
          if (p.res_lp) {
            const u32x2 r = *(const u32x2*)((const uint16_t*)p.res_lp + m * p.ldr_lp + nb);
            v[0] += lo_f<T>(r[0]); v[1] += hi_f<T>(r[0]); v[2] += lo_f<T>(r[1]); v[3] += hi_f<T>(r[1]);
          }
          if (p.res_lp2) {
            const u32x2 r = *(const u32x2*)((const uint16_t*)p.res_lp2 + m * p.ldr_lp2 + nb);
            v[0] += lo_f<T>(r[0]); v[1] += hi_f<T>(r[0]); v[2] += lo_f<T>(r[1]); v[3] += hi_f<T>(r[1]);
          }
          if (p.out_f32) *(float4v*)(p.out_f32 + m * p.ldo_f32 + nb) = v;
          if (p.out_lp) {
            u32x2 o;
            o[0] = pack2<T>(v[0], v[1]);
            o[1] = pack2<T>(v[2], v[3]);
            *(u32x2*)((uint16_t*)p.out_lp + m * p.ldo_lp + nb) = o;
          }
        } else {  // CONVT scatter (pixel shuffle): n = (dy*s + dx)*cout + co
          const int tap = nb / p.ct_cout;
          const int co = nb - tap * p.ct_cout;
          const int dy = tap / p.ct_s, dx = tap - dy * p.ct_s;
          const int64_t pix = ct_base + (int64_t)dy * ((int64_t)p.ct_w * p.ct_s) + dx;
          u32x2 o;
          o[0] = pack2<T>(v[0], v[1]);
          o[1] = pack2<T>(v[2], v[3]);
          *(u32x2*)((uint16_t*)p.out_lp + pix * p.ct_cout + co) = o;
        }
      }
    }
  } else {  // ------------------------------------------------------------ QKV
    const int part = n0 / Dm;  // 0 q, 1 k, 2 v (block-uniform: Dm % 128 == 0)
    if (part < 2) {
      uint16_t* dst = (uint16_t*)(part == 0 ? p.q : p.k);
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) {
        const int64_t m = m0 + wm * 64 + mf * 16 + fr;
        if (m >= p.M) continue;
        float4v v[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
          const int nb = n0 + wn * 64 + nf * 16 + fg * 4;
          v[nf] = acc[nf][mf];
          if (p.bias) v[nf] += *(const float4v*)(p.bias + nb);
        }
        if (p.rope_cos) {
          // RoPE-2D (pos_embed.py:162-183): the wave's 64 columns are one head; dims [0,32) rotate by the
          // row position y, [32,64) by the column position x; dim i pairs with i+16 inside each half.
          const int pos = (int)(m % p.seq_len);
          const int py = pos / p.rope_w, px = pos - py * p.rope_w;
          const int64_t grp = m / p.rope_w;  // rope_mode 1: one angle set per row group (LlamaDecoder: per view)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int64_t toff = p.rope_mode == 1 ? grp * 32 + h * 16 : (int64_t)(h == 0 ? py : px) * 16;
            const float4v c = *(const float4v*)(p.rope_cos + toff + fg * 4);
            const float4v s = *(const float4v*)(p.rope_sin + toff + fg * 4);
            const float4v a = v[2 * h], b = v[2 * h + 1];
            v[2 * h] = a * c - b * s;
            v[2 * h + 1] = b * c + a * s;
          }
        }
        if (part == 0 && p.q_scale != 0.f) {
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) v[nf] *= p.q_scale;
        }
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
          const int nb = n0 + wn * 64 + nf * 16 + fg * 4 - part * Dm;
          u32x2 o;
          o[0] = pack2<T>(v[nf][0], v[nf][1]);
          o[1] = pack2<T>(v[nf][2], v[nf][3]);
          *(u32x2*)(dst + m * (int64_t)Dm + nb) = o;
        }
      }
    } else {
      // V, swapped roles: acc[mf][nf], lane = (col n = fr, rows m = fg*4 + j)
      uint16_t* vt = (uint16_t*)p.vt;
      const bool vec_ok = ((p.seq_len | p.ldvt) & 3) == 0;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        const int n = n0 + wn * 64 + nf * 16 + fr;  // < N (Dm % 128 == 0)
        const int d = n - 2 * Dm;
        const float bb = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
          const int64_t mb = m0 + wm * 64 + mf * 16 + fg * 4;
          if (mb >= p.M) continue;
          const float4v v = acc[mf][nf] + bb;
          if (vec_ok) {  // seq_len % 4 == 0 -> the 4 tokens share a sequence; M % 4 == 0 follows
            const int64_t s = mb / p.seq_len, t = mb % p.seq_len;
            u32x2 o;
            o[0] = pack2<T>(v[0], v[1]);
            o[1] = pack2<T>(v[2], v[3]);
            *(u32x2*)(vt + (s * Dm + d) * p.ldvt + t) = o;
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
  }
}

int g_gemm_glds = -1;  // -1: read F3R_GEMM_GLDS once: 0 register staging, 1 DMA double buffer (BK 64), 2 DMA 4-stage ring (BK 32)
int glds_mode() {
  if (g_gemm_glds < 0) {
    const char* e = getenv("F3R_GEMM_GLDS");
    g_gemm_glds = e ? atoi(e) : 1;
  }
  return g_gemm_glds;
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
  if (a.a_mode == F3R_A_CONV3X3) {
    F3R_REQUIRE(a.epi == F3R_EPI_GENERIC, "f3r_gemm: conv3x3 supports only the generic epilogue");
    return launch<T, F3R_A_CONV3X3, F3R_EPI_GENERIC, 0>(a, stream);
  }
  const int glds = (a.K == a.Kpad && a.M > 0) ? glds_mode() : 0;
  switch (a.epi) {
    case F3R_EPI_GENERIC:
      return glds == 2 ? launch<T, F3R_A_PLAIN, F3R_EPI_GENERIC, 2>(a, stream)
           : glds == 1 ? launch<T, F3R_A_PLAIN, F3R_EPI_GENERIC, 1>(a, stream) : launch<T, F3R_A_PLAIN, F3R_EPI_GENERIC, 0>(a, stream);
    case F3R_EPI_QKV:
      return glds == 2 ? launch<T, F3R_A_PLAIN, F3R_EPI_QKV, 2>(a, stream)
           : glds == 1 ? launch<T, F3R_A_PLAIN, F3R_EPI_QKV, 1>(a, stream) : launch<T, F3R_A_PLAIN, F3R_EPI_QKV, 0>(a, stream);
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
  F3R_REQUIRE(a.Kpad > 0 && a.Kpad % 64 == 0, "f3r_gemm: Kpad %d must be a positive multiple of 64", a.Kpad);
  F3R_REQUIRE(a.N % 4 == 0, "f3r_gemm: N %d must be a multiple of 4", a.N);
  F3R_REQUIRE(al16(a.A) && al16(a.W), "f3r_gemm: A/W must be 16-byte aligned");
  F3R_REQUIRE(a.dtype == F3R_F16 || a.dtype == F3R_BF16, "f3r_gemm: bad dtype %d", a.dtype);
  if (a.a_mode == F3R_A_PLAIN) {
    F3R_REQUIRE(a.K > 0 && a.K % 8 == 0 && a.K <= a.Kpad, "f3r_gemm: K %d must be a multiple of 8 and <= Kpad", a.K);
    F3R_REQUIRE(a.lda % 8 == 0 && a.lda >= a.K, "f3r_gemm: lda %lld must be a multiple of 8 and >= K", (long long)a.lda);
  } else if (a.a_mode == F3R_A_CONV3X3) {
    F3R_REQUIRE(a.conv_C > 0 && a.conv_C % 8 == 0, "f3r_gemm: conv_C %d must be a multiple of 8", a.conv_C);
    F3R_REQUIRE(a.Kpad == 9 * ((a.conv_C + 63) / 64) * 64, "f3r_gemm: conv Kpad %d != 9*roundup(C,64)", a.Kpad);
    F3R_REQUIRE(a.conv_stride == 1 || a.conv_stride == 2, "f3r_gemm: conv stride %d", a.conv_stride);
    F3R_REQUIRE(a.conv_OH == (a.conv_H + 2 - 3) / a.conv_stride + 1 && a.conv_OW == (a.conv_W + 2 - 3) / a.conv_stride + 1,
                "f3r_gemm: conv output dims inconsistent");
    F3R_REQUIRE(a.conv_OH > 0 && a.conv_OW > 0 && a.M % ((int64_t)a.conv_OH * a.conv_OW) == 0, "f3r_gemm: conv M not a multiple of OH*OW");
  } else {
    f3r_set_error("f3r_gemm: bad a_mode %d", a.a_mode);
    return F3R_ERR_ARG;
  }
  if (a.epi == F3R_EPI_GENERIC) {
    F3R_REQUIRE(a.out_f32 || a.out_lp, "f3r_gemm: no output");
    F3R_REQUIRE(!a.out_f32 || (al16(a.out_f32) && a.ldo_f32 % 4 == 0 && a.ldo_f32 >= a.N), "f3r_gemm: out_f32 alignment/ld");
    F3R_REQUIRE(!a.out_lp || (al8(a.out_lp) && a.ldo_lp % 4 == 0 && a.ldo_lp >= a.N), "f3r_gemm: out_lp alignment/ld");
    F3R_REQUIRE(!a.res_f32 || (al16(a.res_f32) && a.ldr_f32 % 4 == 0), "f3r_gemm: res_f32 alignment/ld");
    F3R_REQUIRE(!a.res_lp || (al8(a.res_lp) && a.ldr_lp % 4 == 0), "f3r_gemm: res_lp alignment/ld");
    F3R_REQUIRE(!a.res_lp2 || (al8(a.res_lp2) && a.ldr_lp2 % 4 == 0), "f3r_gemm: res_lp2 alignment/ld");
    F3R_REQUIRE(!a.rowadd || (al16(a.rowadd) && a.rowadd_div > 0), "f3r_gemm: rowadd alignment/div");
    F3R_REQUIRE(!a.bias || al16(a.bias), "f3r_gemm: bias alignment");
  } else if (a.epi == F3R_EPI_QKV) {
    F3R_REQUIRE(a.N % 3 == 0 && (a.N / 3) % 128 == 0, "f3r_gemm: QKV needs N = 3*D with D %% 128 == 0 (got %d)", a.N);
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
  return a.dtype == F3R_F16 ? dispatch<F16>(a, s) : dispatch<BF16>(a, s);
}
