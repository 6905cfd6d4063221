// precision = "exact": the two kernels that let `inference(dtype="32")` mean what it means in the reference (inference_multiview.py:41-52:
// no autocast, fp32 everywhere).  Every GEMM / conv of that mode runs with BOTH operands as hi + lo planes (split "x3": ~22 significand
// bits each, fp32 accumulation), which needs the attention inputs and output in more than 16 bits too -- so the attention itself runs in
// plain fp32 here, on q / k / v taken from an fp32 [T][3D] buffer (the QKV projection through the generic epilogue).  A validation mode for
// scenes of tens of views (one FMA pipe, no MFMA: ~30 TFLOP/s), not a throughput mode.
#include <math.h>

#include "f3r_common.h"

namespace {

// RoPE-2D (croco/models/pos_embed.py:162-183) in place on the q and k parts of qkv[T][ld]: heads 0 .. 2H-1 are the 64-wide column groups
// [0, 2*H*64); dims [0,32) of a head rotate by the token's row position y, [32,64) by its column position x; pairs (i, i + 16).
__global__ void rope2d_f32_kernel(float* __restrict__ qkv, int64_t rows, int64_t ld, int heads2, int64_t seq_len, int rope_w,
                                  const float* __restrict__ cs, const float* __restrict__ sn, int64_t n) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int i = (int)(idx & 15);
  const int half = (int)((idx >> 4) & 1);
  const int64_t rest = idx >> 5;
  const int head = (int)(rest % heads2);
  const int64_t row = rest / heads2;
  const int pos = (int)(row % seq_len);
  const int py = pos / rope_w, px = pos - py * rope_w;
  const int coord = half == 0 ? py : px;
  const float c = cs[(int64_t)coord * 16 + i], s = sn[(int64_t)coord * 16 + i];
  float* p = qkv + row * ld + head * 64 + half * 32 + i;
  const float a = p[0], b = p[16];
  p[0] = a * c - b * s;
  p[16] = b * c + a * s;
}

// softmax(q k^T * scale) v in fp32.  One thread = one query row of one head (q and the output row in registers), one 256-thread workgroup
// = 256 consecutive queries of ONE sequence; 32-key tiles of K and V go through LDS (every lane reads the same address: broadcasts).
constexpr int XQ = 256, XK = 32;
template <class T, int HD>
__global__ __launch_bounds__(XQ) void attn_f32_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int64_t ld,
                                                      uint16_t* __restrict__ o_hi, uint16_t* __restrict__ o_lo, float* __restrict__ o_f32, int64_t ldo,
                                                      int64_t seq_len, int qblocks, float scale) {
  __shared__ __attribute__((aligned(16))) float Ks[XK][HD];
  __shared__ __attribute__((aligned(16))) float Vs[XK][HD];
  const int head = blockIdx.y;
  const int64_t seq = blockIdx.x / qblocks;
  const int64_t qi = (int64_t)(blockIdx.x % qblocks) * XQ + threadIdx.x;  // query index inside the sequence
  const bool q_ok = qi < seq_len;
  const int64_t row0 = seq * seq_len;
  float qr[HD], o[HD];
  {
    const float* src = q + (row0 + (q_ok ? qi : seq_len - 1)) * ld + head * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const float4v t = *(const float4v*)(src + d);
      qr[d] = t[0] * scale; qr[d + 1] = t[1] * scale; qr[d + 2] = t[2] * scale; qr[d + 3] = t[3] * scale;
    }
  }
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
  float m = -INFINITY, l = 0.f;
  for (int64_t k0 = 0; k0 < seq_len; k0 += XK) {
    __syncthreads();  // the previous tile is consumed
    {
      // 32 keys x HD dims x 2 tensors as float4
      for (int e = threadIdx.x; e < XK * (HD / 4); e += XQ) {
        const int kj = e / (HD / 4), c4 = (e % (HD / 4)) * 4;
        const int64_t kr = k0 + kj < seq_len ? k0 + kj : seq_len - 1;
        *(float4v*)&Ks[kj][c4] = *(const float4v*)(k + (row0 + kr) * ld + head * HD + c4);
        *(float4v*)&Vs[kj][c4] = *(const float4v*)(v + (row0 + kr) * ld + head * HD + c4);
      }
    }
    __syncthreads();
    const int valid = (int)(seq_len - k0 < XK ? seq_len - k0 : XK);
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
    const float m_new = fmaxf(m, tmax);  // finite: every tile holds at least one valid key
    const float alpha = expf(m - m_new);  // exp(-inf) = 0 on the first tile
    l *= alpha;
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] *= alpha;
#pragma unroll
    for (int j = 0; j < XK; ++j) {
      const float p = expf(s[j] - m_new);  // 0 for the masked tail
      l += p;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        const float4v vv = *(const float4v*)&Vs[j][d];
        o[d] = fmaf(p, vv[0], o[d]); o[d + 1] = fmaf(p, vv[1], o[d + 1]); o[d + 2] = fmaf(p, vv[2], o[d + 2]); o[d + 3] = fmaf(p, vv[3], o[d + 3]);
      }
    }
    m = m_new;
  }
  if (!q_ok) return;
  const float inv = 1.0f / l;
  const int64_t orow = (row0 + qi) * ldo + head * HD;
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    const float4v r = {o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
    if (o_f32) *(float4v*)(o_f32 + orow + d) = r;
    if (o_hi) {
      u32x2 hi;
      hi[0] = pack2<T>(r[0], r[1]);
      hi[1] = pack2<T>(r[2], r[3]);
      *(u32x2*)(o_hi + orow + d) = hi;
      if (o_lo) {
        u32x2 lo;
        lo[0] = pack2<T>(r[0] - lo_f<T>(hi[0]), r[1] - hi_f<T>(hi[0]));
        lo[1] = pack2<T>(r[2] - lo_f<T>(hi[1]), r[3] - hi_f<T>(hi[1]));
        *(u32x2*)(o_lo + orow + d) = lo;
      }
    }
  }
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int f3r_rope2d_f32(float* qkv, int64_t rows, int64_t ld, int n_heads, int64_t seq_len, int rope_w, const float* rope_cos,
                              const float* rope_sin, f3r_stream_t stream) {
  F3R_REQUIRE(qkv && rope_cos && rope_sin, "f3r_rope2d_f32: null pointer");
  F3R_REQUIRE(rows >= 0 && n_heads > 0 && ld >= (int64_t)2 * n_heads * 64 && seq_len > 0 && rows % seq_len == 0 && rope_w > 0,
              "f3r_rope2d_f32: bad sizes (rows %lld, ld %lld, heads %d, seq_len %lld, rope_w %d)", (long long)rows, (long long)ld, n_heads,
              (long long)seq_len, rope_w);
  const int64_t n = rows * (2 * n_heads) * 32;
  if (n == 0) return F3R_OK;
  F3R_REQUIRE((n + 255) / 256 < (1ll << 31), "f3r_rope2d_f32: grid too large");
  hipLaunchKernelGGL(rope2d_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, qkv, rows, ld, 2 * n_heads, seq_len, rope_w,
                     rope_cos, rope_sin, n);
  return f3r_check_launch("f3r_rope2d_f32");
}

extern "C" int f3r_attn_f32(const float* q, const float* k, const float* v, int64_t ld, void* o_hi, void* o_lo, float* o_f32, int64_t ldo,
                            int64_t n_seq, int64_t seq_len, int n_heads, float scale, int dtype, int head_dim, f3r_stream_t stream) {
  const int hd = head_dim == 0 ? 64 : head_dim;
  F3R_REQUIRE(hd == 16 || hd == 32 || hd == 48 || hd == 64 || hd == 80 || hd == 96 || hd == 112 || hd == 128, "f3r_attn_f32: head_dim %d", head_dim);
  F3R_REQUIRE(q && k && v && (o_hi || o_f32), "f3r_attn_f32: null pointer");
  F3R_REQUIRE(dtype == F3R_F16 || dtype == F3R_BF16, "f3r_attn_f32: bad dtype %d", dtype);
  F3R_REQUIRE(al16(q) && al16(k) && al16(v) && ld % 4 == 0 && ldo % 4 == 0 && (!o_f32 || al16(o_f32)) && ((((uintptr_t)o_hi) | ((uintptr_t)o_lo)) & 7) == 0,
              "f3r_attn_f32: alignment (rows of q / k / v / o_f32 16-byte, o_hi / o_lo 8-byte, strides multiples of 4)");
  F3R_REQUIRE(!o_lo || o_hi, "f3r_attn_f32: a low plane needs its high plane");
  F3R_REQUIRE(n_seq >= 0 && seq_len > 0 && n_heads > 0 && n_heads < 65536 && ld >= (int64_t)n_heads * hd && ldo >= (int64_t)n_heads * hd, "f3r_attn_f32: bad sizes");
  if (n_seq == 0) return F3R_OK;
  const int64_t qblocks = (seq_len + XQ - 1) / XQ;
  F3R_REQUIRE(qblocks * n_seq < (1ll << 31), "f3r_attn_f32: grid too large");
  const dim3 grid((unsigned)(qblocks * n_seq), (unsigned)n_heads);
#define F3R_X(HDV)                                                                                                                              \
  case HDV:                                                                                                                                     \
    if (dtype == F3R_F16)                                                                                                                       \
      hipLaunchKernelGGL((attn_f32_kernel<F16, HDV>), grid, dim3(XQ), 0, (hipStream_t)stream, q, k, v, ld, (uint16_t*)o_hi, (uint16_t*)o_lo, o_f32,   \
                         ldo, seq_len, (int)qblocks, scale);                                                                                    \
    else                                                                                                                                        \
      hipLaunchKernelGGL((attn_f32_kernel<BF16, HDV>), grid, dim3(XQ), 0, (hipStream_t)stream, q, k, v, ld, (uint16_t*)o_hi, (uint16_t*)o_lo, o_f32,  \
                         ldo, seq_len, (int)qblocks, scale);                                                                                    \
    break;
  switch (hd) {
    F3R_X(16) F3R_X(32) F3R_X(48) F3R_X(64) F3R_X(80) F3R_X(96) F3R_X(112) F3R_X(128)
    default: break;
  }
#undef F3R_X
  return f3r_check_launch("f3r_attn_f32");
}
