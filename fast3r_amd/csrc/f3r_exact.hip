// precision = "exact": the two kernels that let `inference(dtype="32")` mean what it means in the reference (inference_multiview.py:41-52:
// no autocast, fp32 everywhere).  Every GEMM / conv of that mode runs with BOTH operands as hi + lo planes (split "x3": ~22 significand
// bits each, fp32 accumulation), which needs the attention inputs and output in more than 16 bits too -- so the attention itself runs in
// plain fp32 here, on q / k / v taken from an fp32 [T][3D] buffer (the QKV projection through the generic epilogue).  A validation mode for
// scenes of tens of views (one FMA pipe, no MFMA: ~30 TFLOP/s), not a throughput mode.
#include <math.h>
#include <string.h>

#include "f3r_common.h"

namespace {

// Rotary embedding in place on the first n_rot 64-wide column groups of qkv[T][ld] (the q heads followed by the k heads).
// mode 0 = RoPE-2D (croco/models/pos_embed.py:162-183): dims [0,32) of a head rotate by the token's row position y, [32,64) by its column
// position x; pairs (i, i + 16); tables [n_pos][16].  mode 1 = the LlamaDecoder form of the QKV epilogue (f3r_gemm_args.rope_mode = 1,
// llama.py:96-122 with the q / k weight rows permuted at pack time): one [32]-angle row per group of rope_w rows, dims [0,32) use table
// columns 0-15, dims [32,64) columns 16-31.
__global__ void rope_f32_kernel(float* __restrict__ qkv, int64_t rows, int64_t ld, int n_rot, int64_t seq_len, int rope_w, int mode,
                                const float* __restrict__ cs, const float* __restrict__ sn, int64_t n) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int i = (int)(idx & 15);
  const int half = (int)((idx >> 4) & 1);
  const int64_t rest = idx >> 5;
  const int head = (int)(rest % n_rot);
  const int64_t row = rest / n_rot;
  int64_t toff;
  if (mode == 1) {
    toff = (row / rope_w) * 32 + half * 16 + i;
  } else {
    const int pos = (int)(row % seq_len);
    const int py = pos / rope_w, px = pos - py * rope_w;
    toff = (int64_t)(half == 0 ? py : px) * 16 + i;
  }
  const float c = cs[toff], s = sn[toff];
  float* p = qkv + row * ld + head * 64 + half * 32 + i;
  const float a = p[0], b = p[16];
  p[0] = a * c - b * s;
  p[16] = b * c + a * s;
}

// SwiGLU gate of FeedForward.forward (llama.py:284) in fp32: out = silu(ab[r][j]) * ab[r][hidden + j] as hi + lo planes (the A operand of
// the X3 down-projection).
template <class T>
__global__ void silu_mul_f32_kernel(const float* __restrict__ ab, uint16_t* __restrict__ o_hi, uint16_t* __restrict__ o_lo, int64_t rows, int hidden) {
  const int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (idx >= rows * hidden) return;
  const int64_t r = idx / hidden;
  const int j = (int)(idx - r * hidden);
  const float4v a = *(const float4v*)(ab + r * 2 * hidden + j);
  const float4v b = *(const float4v*)(ab + r * 2 * hidden + hidden + j);
  float4v v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = a[e] / (1.0f + expf(-a[e])) * b[e];
  u32x2 hi, lo;
  hi[0] = pack2<T>(v[0], v[1]);
  hi[1] = pack2<T>(v[2], v[3]);
  lo[0] = pack2<T>(v[0] - lo_f<T>(hi[0]), v[1] - hi_f<T>(hi[0]));
  lo[1] = pack2<T>(v[2] - lo_f<T>(hi[1]), v[3] - hi_f<T>(hi[1]));
  *(u32x2*)(o_hi + idx) = hi;
  *(u32x2*)(o_lo + idx) = lo;
}

// softmax(q k^T * scale) v in fp32.  One thread = one query row of one head (q and the output row in registers), one 256-thread workgroup
// = 256 consecutive queries of ONE sequence; 32-key tiles of K and V go through LDS (every lane reads the same address: broadcasts).
// Query head h reads K / V head h / kv_group (repeat_kv, llama.py:195-198); causal: key j of the sequence is visible to query i iff
// k_pos0 + j <= q_pos0 + i (absolute token positions, so the mask composes with view sharding).
constexpr int XQ = 256, XK = 32;
struct AttnF32 {
  const float *q, *k, *v;
  int64_t ldq, ldkv;
  uint16_t *o_hi, *o_lo;
  float* o_f32;
  int64_t ldo, tq, tk, q_pos0, k_pos0;
  int qblocks, kv_group, causal;
  float scale;
};
template <class T, int HD>
__global__ __launch_bounds__(XQ) void attn_f32_kernel(const AttnF32 p) {
  __shared__ __attribute__((aligned(16))) float Ks[XK][HD];
  __shared__ __attribute__((aligned(16))) float Vs[XK][HD];
  const int head = blockIdx.y;
  const int kvh = head / p.kv_group;
  const int64_t seq = blockIdx.x / p.qblocks;
  const int64_t q0 = (int64_t)(blockIdx.x % p.qblocks) * XQ;
  const int64_t qi = q0 + threadIdx.x;  // query index inside the sequence
  const bool q_ok = qi < p.tq;
  const int64_t qrow0 = seq * p.tq, krow0 = seq * p.tk;
  float qr[HD], o[HD];
  {
    const float* src = p.q + (qrow0 + (q_ok ? qi : p.tq - 1)) * p.ldq + head * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const float4v t = *(const float4v*)(src + d);
      qr[d] = t[0] * p.scale; qr[d + 1] = t[1] * p.scale; qr[d + 2] = t[2] * p.scale; qr[d + 3] = t[3] * p.scale;
    }
  }
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
  float m = -INFINITY, l = 0.f;
  // causal: keys past the workgroup's last query are invisible to all of its rows (workgroup-uniform bound)
  int64_t k_end = p.tk;
  if (p.causal) {
    const int64_t last_q = (q0 + XQ < p.tq ? q0 + XQ : p.tq) - 1;
    const int64_t vis = p.q_pos0 + last_q - p.k_pos0 + 1;
    k_end = vis < 0 ? 0 : (vis < p.tk ? vis : p.tk);
  }
  const int64_t my_vis = p.causal ? p.q_pos0 + qi - p.k_pos0 + 1 : p.tk;  // keys [0, my_vis) are visible to this row
  for (int64_t k0 = 0; k0 < k_end; k0 += XK) {
    __syncthreads();  // the previous tile is consumed
    for (int e = threadIdx.x; e < XK * (HD / 4); e += XQ) {  // 32 keys x HD dims x 2 tensors as float4
      const int kj = e / (HD / 4), c4 = (e % (HD / 4)) * 4;
      const int64_t kr = k0 + kj < p.tk ? k0 + kj : p.tk - 1;
      *(float4v*)&Ks[kj][c4] = *(const float4v*)(p.k + (krow0 + kr) * p.ldkv + kvh * HD + c4);
      *(float4v*)&Vs[kj][c4] = *(const float4v*)(p.v + (krow0 + kr) * p.ldkv + kvh * HD + c4);
    }
    __syncthreads();
    int64_t lim = p.tk < my_vis ? p.tk : my_vis;
    const int valid = (int)(lim - k0 < XK ? (lim - k0 < 0 ? 0 : lim - k0) : XK);
    float s[XK];
    float tmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < XK; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        const float4v kk = *(const float4v*)&Ks[j][d];
        acc = fmaf(qr[d], kk[0], acc); acc = fmaf(qr[d + 1], kk[1], acc); acc = fmaf(qr[d + 2], kk[2], acc); acc = fmaf(qr[d + 3], kk[3], acc);
      }
      s[j] = j < valid ? acc : -INFINITY;
      tmax = fmaxf(tmax, s[j]);
    }
    if (valid <= 0) continue;  // (causal) nothing of this tile is visible to this row; the barriers above are workgroup-uniform
    const float m_new = fmaxf(m, tmax);  // finite: the tile holds at least one visible key
    const float alpha = expf(m - m_new);  // exp(-inf) = 0 on the first tile
    l *= alpha;
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] *= alpha;
#pragma unroll
    for (int j = 0; j < XK; ++j) {
      const float pj = expf(s[j] - m_new);  // 0 for the masked tail
      l += pj;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        const float4v vv = *(const float4v*)&Vs[j][d];
        o[d] = fmaf(pj, vv[0], o[d]); o[d + 1] = fmaf(pj, vv[1], o[d + 1]); o[d + 2] = fmaf(pj, vv[2], o[d + 2]); o[d + 3] = fmaf(pj, vv[3], o[d + 3]);
      }
    }
    m = m_new;
  }
  if (!q_ok) return;
  const float inv = l > 0.f ? 1.0f / l : 0.f;  // a row that sees no key at all (causal with k_pos0 > its position) yields zeros
  const int64_t orow = (qrow0 + qi) * p.ldo + head * HD;
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    const float4v r = {o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
    if (p.o_f32) *(float4v*)(p.o_f32 + orow + d) = r;
    if (p.o_hi) {
      u32x2 hi;
      hi[0] = pack2<T>(r[0], r[1]);
      hi[1] = pack2<T>(r[2], r[3]);
      *(u32x2*)(p.o_hi + orow + d) = hi;
      if (p.o_lo) {
        u32x2 lo;
        lo[0] = pack2<T>(r[0] - lo_f<T>(hi[0]), r[1] - hi_f<T>(hi[0]));
        lo[1] = pack2<T>(r[2] - lo_f<T>(hi[1]), r[3] - hi_f<T>(hi[1]));
        *(u32x2*)(p.o_lo + orow + d) = lo;
      }
    }
  }
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int f3r_rope_f32(float* qkv, int64_t rows, int64_t ld, int n_rot_heads, int64_t seq_len, int rope_w, int rope_mode,
                            const float* rope_cos, const float* rope_sin, f3r_stream_t stream) {
  F3R_REQUIRE(qkv && rope_cos && rope_sin, "f3r_rope_f32: null pointer");
  F3R_REQUIRE(rope_mode == 0 || rope_mode == 1, "f3r_rope_f32: rope_mode %d", rope_mode);
  F3R_REQUIRE(rows >= 0 && n_rot_heads > 0 && ld >= (int64_t)n_rot_heads * 64 && seq_len > 0 && rows % seq_len == 0 && rope_w > 0,
              "f3r_rope_f32: bad sizes (rows %lld, ld %lld, rotated heads %d, seq_len %lld, rope_w %d)", (long long)rows, (long long)ld, n_rot_heads,
              (long long)seq_len, rope_w);
  const int64_t n = rows * n_rot_heads * 32;
  if (n == 0) return F3R_OK;
  F3R_REQUIRE((n + 255) / 256 < (1ll << 31), "f3r_rope_f32: grid too large");
  hipLaunchKernelGGL(rope_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, qkv, rows, ld, n_rot_heads, seq_len, rope_w,
                     rope_mode, rope_cos, rope_sin, n);
  return f3r_check_launch("f3r_rope_f32");
}

extern "C" int f3r_rope2d_f32(float* qkv, int64_t rows, int64_t ld, int n_heads, int64_t seq_len, int rope_w, const float* rope_cos,
                              const float* rope_sin, f3r_stream_t stream) {
  return f3r_rope_f32(qkv, rows, ld, 2 * n_heads, seq_len, rope_w, 0, rope_cos, rope_sin, stream);
}

extern "C" int f3r_silu_mul_f32(const float* ab, void* out_hi, void* out_lo, int64_t rows, int hidden, int dtype, f3r_stream_t stream) {
  F3R_REQUIRE(ab && out_hi && out_lo && al16(ab) && ((((uintptr_t)out_hi) | ((uintptr_t)out_lo)) & 7) == 0, "f3r_silu_mul_f32: null / misaligned pointer");
  F3R_REQUIRE(dtype == F3R_F16 || dtype == F3R_BF16, "f3r_silu_mul_f32: bad dtype %d", dtype);
  F3R_REQUIRE(rows >= 0 && hidden > 0 && hidden % 4 == 0, "f3r_silu_mul_f32: hidden %d must be a positive multiple of 4", hidden);
  const int64_t n = rows * hidden / 4;
  if (n == 0) return F3R_OK;
  F3R_REQUIRE((n + 255) / 256 < (1ll << 31), "f3r_silu_mul_f32: grid too large");
  const dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == F3R_F16)
    hipLaunchKernelGGL(silu_mul_f32_kernel<F16>, grid, dim3(256), 0, (hipStream_t)stream, ab, (uint16_t*)out_hi, (uint16_t*)out_lo, rows, hidden);
  else
    hipLaunchKernelGGL(silu_mul_f32_kernel<BF16>, grid, dim3(256), 0, (hipStream_t)stream, ab, (uint16_t*)out_hi, (uint16_t*)out_lo, rows, hidden);
  return f3r_check_launch("f3r_silu_mul_f32");
}

extern "C" int f3r_attn_f32_ex(const f3r_attn_f32_args* args, f3r_stream_t stream) {
  F3R_REQUIRE(args, "f3r_attn_f32_ex: null args");
  const f3r_attn_f32_args& a = *args;
  const int hd = a.head_dim == 0 ? 64 : a.head_dim;
  const int grp = a.kv_group <= 0 ? 1 : a.kv_group;
  F3R_REQUIRE(hd == 16 || hd == 32 || hd == 48 || hd == 64 || hd == 80 || hd == 96 || hd == 112 || hd == 128, "f3r_attn_f32: head_dim %d", a.head_dim);
  F3R_REQUIRE(a.q && a.k && a.v && (a.o_hi || a.o_f32), "f3r_attn_f32: null pointer");
  F3R_REQUIRE(a.dtype == F3R_F16 || a.dtype == F3R_BF16, "f3r_attn_f32: bad dtype %d", a.dtype);
  F3R_REQUIRE(al16(a.q) && al16(a.k) && al16(a.v) && a.ldq % 4 == 0 && a.ldkv % 4 == 0 && a.ldo % 4 == 0 && (!a.o_f32 || al16(a.o_f32)) &&
                  ((((uintptr_t)a.o_hi) | ((uintptr_t)a.o_lo)) & 7) == 0,
              "f3r_attn_f32: alignment (rows of q / k / v / o_f32 16-byte, o_hi / o_lo 8-byte, strides multiples of 4)");
  F3R_REQUIRE(!a.o_lo || a.o_hi, "f3r_attn_f32: a low plane needs its high plane");
  F3R_REQUIRE(a.n_seq >= 0 && a.tq > 0 && a.tk > 0 && a.n_heads > 0 && a.n_heads < 65536 && a.n_heads % grp == 0 && a.ldq >= (int64_t)a.n_heads * hd &&
                  a.ldkv >= (int64_t)(a.n_heads / grp) * hd && a.ldo >= (int64_t)a.n_heads * hd,
              "f3r_attn_f32: bad sizes");
  if (a.n_seq == 0) return F3R_OK;
  const int64_t qblocks = (a.tq + XQ - 1) / XQ;
  F3R_REQUIRE(qblocks * a.n_seq < (1ll << 31), "f3r_attn_f32: grid too large");
  const dim3 grid((unsigned)(qblocks * a.n_seq), (unsigned)a.n_heads);
  AttnF32 p;
  p.q = a.q; p.k = a.k; p.v = a.v; p.ldq = a.ldq; p.ldkv = a.ldkv;
  p.o_hi = (uint16_t*)a.o_hi; p.o_lo = (uint16_t*)a.o_lo; p.o_f32 = a.o_f32; p.ldo = a.ldo;
  p.tq = a.tq; p.tk = a.tk; p.q_pos0 = a.q_pos0; p.k_pos0 = a.k_pos0;
  p.qblocks = (int)qblocks; p.kv_group = grp; p.causal = a.causal ? 1 : 0; p.scale = a.scale;
#define F3R_X(HDV)                                                                                                  \
  case HDV:                                                                                                         \
    if (a.dtype == F3R_F16)                                                                                         \
      hipLaunchKernelGGL((attn_f32_kernel<F16, HDV>), grid, dim3(XQ), 0, (hipStream_t)stream, p);                   \
    else                                                                                                            \
      hipLaunchKernelGGL((attn_f32_kernel<BF16, HDV>), grid, dim3(XQ), 0, (hipStream_t)stream, p);                  \
    break;
  switch (hd) {
    F3R_X(16) F3R_X(32) F3R_X(48) F3R_X(64) F3R_X(80) F3R_X(96) F3R_X(112) F3R_X(128)
    default: break;
  }
#undef F3R_X
  return f3r_check_launch("f3r_attn_f32");
}

extern "C" int f3r_attn_f32(const float* q, const float* k, const float* v, int64_t ld, void* o_hi, void* o_lo, float* o_f32, int64_t ldo,
                            int64_t n_seq, int64_t seq_len, int n_heads, float scale, int dtype, int head_dim, f3r_stream_t stream) {
  f3r_attn_f32_args a;
  memset(&a, 0, sizeof(a));
  a.q = q; a.k = k; a.v = v; a.ldq = ld; a.ldkv = ld;
  a.o_hi = o_hi; a.o_lo = o_lo; a.o_f32 = o_f32; a.ldo = ldo;
  a.n_seq = n_seq; a.tq = seq_len; a.tk = seq_len; a.n_heads = n_heads; a.kv_group = 1; a.dtype = dtype; a.head_dim = head_dim; a.scale = scale;
  return f3r_attn_f32_ex(&a, stream);
}
