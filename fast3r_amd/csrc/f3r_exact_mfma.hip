// f3r_attn_f32_mfma: the fp32-equivalent attention of precision "exact" (f3r_attn_f32_ex, f3r_exact.hip: what inference(dtype="32") means in the
// reference, fast3r/dust3r/inference_multiview.py:41-52 + croco/models/blocks.py:158-169) on the matrix pipe, for the sizes where the FMA-pipe
// kernel takes minutes (it runs ~30 TFLOP/s; O(T^2): N = 100 views is 1e15 FLOP).
//
// Same numbers to ~1e-6, different arithmetic: every operand is an exact sum of two 16-bit planes (hi = lp(x), lo = lp(x - hi): ~22
// significand bits, the X3 scheme of f3r_gemm) and every product is three MFMAs with fp32 accumulation -- the lo x lo term (2^-22 relative) is
// dropped:
//     S  = Qh Kh^T + Qh Kl^T + Ql Kh^T         (q pre-multiplied by scale * log2 e in fp32, BEFORE the split)
//     P  = exp2(S - m) in fp32 (plain online softmax: per-tile row maximum, rescale of O and l), row sums over the unrounded P
//     O += Ph V + Ph Vl + Pl V                  (P split into planes after the exponential)
// Three launches: two pre-passes that write the planes into a caller-owned workspace (q and k rows; V transposed to [head_dim planes][keys],
// the A operand of the O^T = V^T P^T product as in f3r_attn.hip), then the attention kernel.  Layouts, the swapped Q K^T with the pi row
// order and the accumulator -> operand identity are those of f3r_attn.hip; the schedule is the compiler's (a validation mode: 3x the
// MFMAs of the 16-bit kernel and a softmax that takes a maximum per tile).
//
// The FMA kernel stays the reference implementation of the mode: ops.attention_f32 takes this form only from ATTN_F32_MFMA_MIN_KEYS keys on,
// tests/test_exact_mfma_gpu.py compares the two on the same inputs.  head_dim 64, no causal mask (everything else: the FMA kernel).
#include "f3r_common.h"

namespace {

constexpr int XM_NW = 4;              // waves per workgroup, one 32-query block per wave
constexpr int XM_QB = XM_NW * 32;     // queries per workgroup
constexpr int XM_KB = 64;             // keys per tile
constexpr int XM_TILE = 64 * 64;      // elements of one plane of a K or V^T tile

struct AttnX3 {
  const uint16_t *q_hi, *q_lo;    // [n_seq * tq][n_heads * 64]
  const uint16_t *k_hi, *k_lo;    // [n_seq * tk][kv_heads * 64]
  const uint16_t *vt_hi, *vt_lo;  // [n_seq][kv_heads * 64][ldvt], columns >= tk zero
  uint16_t *o_hi, *o_lo;
  float* o_f32;
  int64_t ldo, tq, tk, ldvt;
  int n_heads, kv_heads, kv_group, qblocks;
};

__device__ __forceinline__ int xswz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

// fp32 rows -> hi + lo planes, scaled: out[r][c] = planes(in[r][c0 + c] * scale), 4 elements per thread
template <class T>
__global__ void split_rows_kernel(const float* __restrict__ in, int64_t ld, int64_t rows, int n, float scale, uint16_t* __restrict__ hi,
                                  uint16_t* __restrict__ lo) {
  const int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (idx >= rows * n) return;
  const int64_t r = idx / n;
  const int c = (int)(idx - r * n);
  float4v v = *(const float4v*)(in + r * ld + c);
  v *= scale;
  u32x2 h, l;
  h[0] = pack2<T>(v[0], v[1]);
  h[1] = pack2<T>(v[2], v[3]);
  l[0] = pack2<T>(v[0] - lo_f<T>(h[0]), v[1] - hi_f<T>(h[0]));
  l[1] = pack2<T>(v[2] - lo_f<T>(h[1]), v[3] - hi_f<T>(h[1]));
  *(u32x2*)(hi + idx) = h;
  *(u32x2*)(lo + idx) = l;
}

// V fp32 [n_seq * tk][ld] (columns c0 .. c0 + n) -> V^T planes [n_seq][n][ldvt]; 64 x 64 tiles through LDS, key columns >= tk written as zeros
template <class T>
__global__ __launch_bounds__(256) void transpose_split_kernel(const float* __restrict__ in, int64_t ld, int64_t tk, int n, int64_t ldvt,
                                                              uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
  __shared__ float tile[64][65];
  const int64_t seq = blockIdx.z;
  const int64_t k0 = (int64_t)blockIdx.x * 64;
  const int d0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {  // row k0 + i of V, columns d0 + tx
    const int64_t key = k0 + i;
    tile[i][tx] = key < tk ? in[(seq * tk + key) * ld + d0 + tx] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {  // row d0 + i of V^T, key k0 + tx
    const float v = tile[tx][i];
    const uint16_t h = to_lp<T>(v);
    const int64_t o = (seq * n + d0 + i) * ldvt + k0 + tx;
    hi[o] = h;
    lo[o] = to_lp<T>(v - from_lp<T>(h));
  }
}

template <class T>
__global__ __launch_bounds__(XM_NW * 64, 2) void attn_x3_kernel(const AttnX3 p) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * 4 * XM_TILE];  // [buf][K hi | K lo | Vt hi | Vt lo][64][64] = 64 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 31, g = lane >> 5;
  const int head = blockIdx.y;
  const int kv_head = head / p.kv_group;
  const int64_t seq = blockIdx.x / p.qblocks;
  const int64_t q0 = (int64_t)(blockIdx.x % p.qblocks) * XM_QB + wid * 32;
  const int Dq = p.n_heads * 64, Dk = p.kv_heads * 64;

  int64_t qi = q0 + lq;
  const bool q_ok = qi < p.tq;
  if (!q_ok) qi = p.tq - 1;
  const int64_t qrow = seq * p.tq + qi;
  typename T::vec8 qh[4], ql[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    qh[ds] = as_vec8<T>(*(const u32x4*)(p.q_hi + qrow * Dq + head * 64 + ds * 16 + g * 8));
    ql[ds] = as_vec8<T>(*(const u32x4*)(p.q_lo + qrow * Dq + head * 64 + ds * 16 + g * 8));
  }

  // ---- tile loads: every thread moves 16-byte chunks of the four planes into the swizzled LDS image (rows past tk re-read the last key:
  // masked in the softmax; V^T columns past tk are zeros in the workspace)
  const uint16_t* Kh = p.k_hi + (seq * p.tk) * Dk + kv_head * 64;
  const uint16_t* Kl = p.k_lo + (seq * p.tk) * Dk + kv_head * 64;
  const uint16_t* Vh = p.vt_hi + (seq * Dk + (int64_t)kv_head * 64) * p.ldvt;
  const uint16_t* Vl = p.vt_lo + (seq * Dk + (int64_t)kv_head * 64) * p.ldvt;
  const int64_t n_tiles = (p.tk + XM_KB - 1) / XM_KB;
  u32x4 stage[8];
  auto load_tile = [&](int64_t t) {  // 512 chunks per plane, 256 threads: 2 chunks per plane and thread
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i;
      const int row = c >> 3, ch = c & 7;
      int64_t key = t * XM_KB + row;
      if (key >= p.tk) key = p.tk - 1;
      stage[i] = *(const u32x4*)(Kh + key * Dk + ch * 8);
      stage[2 + i] = *(const u32x4*)(Kl + key * Dk + ch * 8);
      stage[4 + i] = *(const u32x4*)(Vh + (int64_t)row * p.ldvt + t * XM_KB + ch * 8);
      stage[6 + i] = *(const u32x4*)(Vl + (int64_t)row * p.ldvt + t * XM_KB + ch * 8);
    }
  };
  auto store_tile = [&](int buf) {
    uint16_t* base = lds + buf * 4 * XM_TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 256 * i;
      const int row = c >> 3, ch = c & 7;
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) *(u32x4*)(base + pl * XM_TILE + xswz(row, ch)) = stage[2 * pl + i];
    }
  };

  // Blocked accumulation (round 5): a tile's P V products and row sum start from zero and are added to a GROUP accumulator once per tile, the
  // group (XM_GROUP tiles) to the running sums once per group -- the running O and l see n_tiles / XM_GROUP roundings instead of 12 n_tiles / 64
  // n_tiles (at 327 680 keys the sequential form measured 6.1e-6 against float64 where a plain fp32 softmax has 6.5e-7;
  // tests/test_exact_mfma_gpu.py).  A move of the reference rescales all levels (alpha is exactly 1 otherwise).
  constexpr int XM_GROUP = 32;
  float16v o[2], og[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; og[0][i] = 0.f; og[1][i] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f, l_grp = 0.f;
  const int krow_pi = (lq & 0x13) | ((lq & 4) << 1) | ((lq & 8) >> 1);

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int64_t t = 0; t < n_tiles; ++t) {
    const int cur = (int)(t & 1);
    const bool more = t + 1 < n_tiles;
    if (more) load_tile(t + 1);  // into registers: stored after this tile's math
    const uint16_t* kt = lds + cur * 4 * XM_TILE;
    const uint16_t* vt = kt + 2 * XM_TILE;
    // ---- S^T = K Q^T in exp2 units (three plane products)
    float16v s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        const int off = xswz(kb * 32 + krow_pi, ds * 2 + g);
        const typename T::vec8 ah = as_vec8<T>(*(const u32x4*)(kt + off));
        const typename T::vec8 al = as_vec8<T>(*(const u32x4*)(kt + XM_TILE + off));
        s[kb] = T::mfma32(al, qh[ds], s[kb]);  // small terms first
        s[kb] = T::mfma32(ah, ql[ds], s[kb]);
        s[kb] = T::mfma32(ah, qh[ds], s[kb]);
      }
    }
    // register r of block kb is key  kb*32 + 16*(r>>3) + 8*g + (r&7)  of the tile
    const int64_t valid = p.tk - t * XM_KB;
    if (valid < XM_KB) {
      const int vg = (int)valid - 8 * g;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kc = kb * 32 + 16 * (r >> 3) + (r & 7);
          if (kc >= vg) s[kb][r] = -INFINITY;
        }
    }
    // ---- online softmax in fp32: the row's maximum over this tile (a tile always holds a valid key), rescale, exponentials, planes of P
    float mx = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f(m_run - m_new);  // 0 on the first tile
    m_run = m_new;
    l_run *= alpha;
    l_grp *= alpha;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; og[0][i] *= alpha; og[1][i] *= alpha; }
    float l_tile = 0.f;
    typename T::vec8 ph[4], pl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 hk, lk;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float p0 = exp2f(s[ks >> 1][(ks & 1) * 8 + 2 * j] - m_new);
        const float p1 = exp2f(s[ks >> 1][(ks & 1) * 8 + 2 * j + 1] - m_new);
        l_tile += p0 + p1;
        hk[j] = pack2<T>(p0, p1);
        lk[j] = pack2<T>(p0 - lo_f<T>(hk[j]), p1 - hi_f<T>(hk[j]));
      }
      ph[ks] = as_vec8<T>(hk);
      pl[ks] = as_vec8<T>(lk);
    }
    // ---- O^T += V^T P^T (three plane products, from zero per tile; then tile -> group -> running sums)
    l_grp += l_tile;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      float16v ot;
#pragma unroll
      for (int i = 0; i < 16; ++i) ot[i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int off = xswz(db * 32 + lq, ks * 2 + g);
        const typename T::vec8 vh = as_vec8<T>(*(const u32x4*)(vt + off));
        const typename T::vec8 vl = as_vec8<T>(*(const u32x4*)(vt + XM_TILE + off));
        ot = T::mfma32(vl, ph[ks], ot);
        ot = T::mfma32(vh, pl[ks], ot);
        ot = T::mfma32(vh, ph[ks], ot);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) og[db][i] += ot[i];
    }
    if ((t % XM_GROUP) == XM_GROUP - 1 || !more) {
      l_run += l_grp;
      l_grp = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        o[0][i] += og[0][i];
        o[1][i] += og[1][i];
        og[0][i] = 0.f;
        og[1][i] = 0.f;
      }
    }
    if (more) store_tile(cur ^ 1);  // buffer cur^1 was last read before the barrier that ended the previous iteration
    __syncthreads();
  }

  // ---- normalise and store O[q][head*64 + d], d = db*32 + (r&3) + 8*(r>>2) + 4*g
  if (!q_ok) return;
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  const int64_t orow = qrow * p.ldo + head * 64;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const float4v r = {o[db][rq * 4 + 0] * inv, o[db][rq * 4 + 1] * inv, o[db][rq * 4 + 2] * inv, o[db][rq * 4 + 3] * inv};
      const int64_t at = orow + db * 32 + 8 * rq + 4 * g;
      if (p.o_f32) *(float4v*)(p.o_f32 + at) = r;
      if (p.o_hi) {
        u32x2 hi;
        hi[0] = pack2<T>(r[0], r[1]);
        hi[1] = pack2<T>(r[2], r[3]);
        *(u32x2*)(p.o_hi + at) = hi;
        if (p.o_lo) {
          u32x2 lo;
          lo[0] = pack2<T>(r[0] - lo_f<T>(hi[0]), r[1] - hi_f<T>(hi[0]));
          lo[1] = pack2<T>(r[2] - lo_f<T>(hi[1]), r[3] - hi_f<T>(hi[1]));
          *(u32x2*)(p.o_lo + at) = lo;
        }
      }
    }
}

struct Plan {
  int64_t q_elems, k_elems, vt_elems, ldvt, bytes;
  int kv_heads;
};
// can this launch take the MFMA form, and what does it need?
bool plan_of(const f3r_attn_f32_args& a, Plan* pl) {
  const int hd = a.head_dim == 0 ? 64 : a.head_dim;
  const int kvg = a.kv_group > 1 ? a.kv_group : 1;
  if (hd != 64 || a.causal || a.n_heads <= 0 || a.n_heads % kvg != 0 || a.n_seq <= 0 || a.tq <= 0 || a.tk <= 0) return false;
  pl->kv_heads = a.n_heads / kvg;
  pl->ldvt = (a.tk + 63) / 64 * 64;
  pl->q_elems = a.n_seq * a.tq * (int64_t)a.n_heads * 64;
  pl->k_elems = a.n_seq * a.tk * (int64_t)pl->kv_heads * 64;
  pl->vt_elems = a.n_seq * (int64_t)pl->kv_heads * 64 * pl->ldvt;
  pl->bytes = (pl->q_elems + pl->k_elems + pl->vt_elems) * 2 * 2;  // hi + lo planes of 16-bit elements
  return true;
}

}  // namespace

extern "C" int64_t f3r_attn_f32_mfma_workspace(const f3r_attn_f32_args* args) {
  Plan pl;
  if (!args || !plan_of(*args, &pl)) return 0;
  return pl.bytes;
}

extern "C" int f3r_attn_f32_mfma(const f3r_attn_f32_args* args, void* workspace, int64_t workspace_bytes, f3r_stream_t stream) {
  F3R_REQUIRE(args, "f3r_attn_f32_mfma: null args");
  const f3r_attn_f32_args& a = *args;
  Plan pl;
  if (!plan_of(a, &pl)) {
    f3r_set_error("f3r_attn_f32_mfma: head_dim 64 without a causal mask only (f3r_attn_f32_ex takes everything else)");
    return F3R_ERR_UNSUPPORTED;
  }
  F3R_REQUIRE(a.q && a.k && a.v && (a.o_hi || a.o_f32), "f3r_attn_f32_mfma: null pointer");
  F3R_REQUIRE(a.dtype == F3R_F16 || a.dtype == F3R_BF16, "f3r_attn_f32_mfma: bad dtype %d", a.dtype);
  F3R_REQUIRE(!a.o_lo || a.o_hi, "f3r_attn_f32_mfma: a low plane needs its high plane");
  F3R_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= pl.bytes, "f3r_attn_f32_mfma: workspace of %lld bytes (16-byte aligned) needed",
              (long long)pl.bytes);
  const int Dq = a.n_heads * 64, Dk = pl.kv_heads * 64;
  F3R_REQUIRE((((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.o_f32) & 15) == 0 && (((uintptr_t)a.o_hi | (uintptr_t)a.o_lo) & 7) == 0 &&
                  a.ldq % 4 == 0 && a.ldkv % 4 == 0 && a.ldo % 4 == 0 && a.ldq >= Dq && a.ldkv >= Dk && a.ldo >= Dq,
              "f3r_attn_f32_mfma: alignment / strides");
  const int64_t qblocks = (a.tq + XM_QB - 1) / XM_QB;
  F3R_REQUIRE(qblocks * a.n_seq < (1ll << 31) && a.n_heads < 65536 && a.n_seq < 65536 && (pl.ldvt / 64) < (1ll << 31), "f3r_attn_f32_mfma: grid too large");
  uint16_t* w = (uint16_t*)workspace;
  uint16_t* q_hi = w;
  uint16_t* q_lo = q_hi + pl.q_elems;
  uint16_t* k_hi = q_lo + pl.q_elems;
  uint16_t* k_lo = k_hi + pl.k_elems;
  uint16_t* vt_hi = k_lo + pl.k_elems;
  uint16_t* vt_lo = vt_hi + pl.vt_elems;
  hipStream_t s = (hipStream_t)stream;
  const float cq = a.scale * 1.44269504088896340736f;
  const int64_t qrows = a.n_seq * a.tq, krows = a.n_seq * a.tk;
  const dim3 gq((unsigned)((qrows * Dq / 4 + 255) / 256)), gk((unsigned)((krows * Dk / 4 + 255) / 256));
  const dim3 gv((unsigned)(pl.ldvt / 64), (unsigned)(Dk / 64), (unsigned)a.n_seq);
  AttnX3 p;
  p.q_hi = q_hi; p.q_lo = q_lo; p.k_hi = k_hi; p.k_lo = k_lo; p.vt_hi = vt_hi; p.vt_lo = vt_lo;
  p.o_hi = (uint16_t*)a.o_hi; p.o_lo = (uint16_t*)a.o_lo; p.o_f32 = a.o_f32;
  p.ldo = a.ldo; p.tq = a.tq; p.tk = a.tk; p.ldvt = pl.ldvt;
  p.n_heads = a.n_heads; p.kv_heads = pl.kv_heads; p.kv_group = a.n_heads / pl.kv_heads; p.qblocks = (int)qblocks;
  const dim3 ga((unsigned)(qblocks * a.n_seq), (unsigned)a.n_heads);
#define F3R_X3(TT)                                                                                                                  \
  hipLaunchKernelGGL(split_rows_kernel<TT>, gq, dim3(256), 0, s, a.q, a.ldq, qrows, Dq, cq, q_hi, q_lo);                            \
  hipLaunchKernelGGL(split_rows_kernel<TT>, gk, dim3(256), 0, s, a.k, a.ldkv, krows, Dk, 1.0f, k_hi, k_lo);                         \
  hipLaunchKernelGGL(transpose_split_kernel<TT>, gv, dim3(256), 0, s, a.v, a.ldkv, a.tk, Dk, pl.ldvt, vt_hi, vt_lo);                \
  hipLaunchKernelGGL(attn_x3_kernel<TT>, ga, dim3(XM_NW * 64), 0, s, p)
  if (a.dtype == F3R_F16) { F3R_X3(F16); } else { F3R_X3(BF16); }
#undef F3R_X3
  return f3r_check_launch("f3r_attn_f32_mfma");
}
