// precision "robust" (round 6): the two small passes around the three-product attention kernel (csrc/asm/attn_gen.py, AttnGen(qk_planes = 2)).
//
// The attention core of Attention.forward (croco/models/blocks.py:158-190) with Q and K carried as hi + lo fp16 planes: S = q_hi k_hi + q_lo k_hi +
// q_hi k_lo (fp32 accumulate), P and V single fp16 -- the cheapest operand set that keeps a noise-amplifying checkpoint within 1e-3 of the fp32
// path (oracle/precision_study.py --study robust_vitl: ViT-L with heavy-tailed weights 3.6e-4, against 2.5e-3 for one fp16 product everywhere).
//
//   f3r_qkv_planes         fp32 [rows][q | k | v] (the X3 QKV projection's output, rotary embedding already applied) ->
//                            q rows [rows][heads][hi 64 | lo 64] fp16, pre-multiplied by scale * log2(e) BEFORE the split,
//                            k rows [rows][kv_heads][hi 64 | lo 64],
//                            V^T    [n_seq][kv_heads * 64][ldvt] fp16 (one plane; key columns >= seq_len are written as zeros)
//   f3r_attn_state_finish  the parked online-softmax state of the attention launch (f3r_attn_args.state_out: un-normalised O fp32, {m, l0, l1})
//                          -> O / (l0 + l1) as hi + lo planes (the A operand of the X3 output projection) and / or fp32
#include "f3r_common.h"

namespace {

// q / k part: one thread per 4 consecutive columns of a head
template <class T, bool F8>
__global__ void planes_rows_kernel(const float* __restrict__ in, int64_t ld, int64_t rows, int n_heads, float scale, uint16_t* __restrict__ out) {
  const int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;   // element index in [rows][n_heads * 64]
  const int64_t n = (int64_t)n_heads * 64;
  if (idx >= rows * n) return;
  const int64_t r = idx / n;
  const int c = (int)(idx - r * n);
  const int h = c >> 6, d = c & 63;
  float4v v = *(const float4v*)(in + r * ld + c);
  v *= scale;
  u32x2 hi, lo;
  hi[0] = pack2<T>(v[0], v[1]);
  hi[1] = pack2<T>(v[2], v[3]);
  lo[0] = pack2<T>(v[0] - lo_f<T>(hi[0]), v[1] - hi_f<T>(hi[0]));
  lo[1] = pack2<T>(v[2] - lo_f<T>(hi[1]), v[3] - hi_f<T>(hi[1]));
  uint16_t* o = out + (r * n_heads + h) * 128 + d;
  *(u32x2*)o = hi;
  if constexpr (!F8) {
    *(u32x2*)(o + 64) = lo;
  } else {
    // planes mode 3: bytes [128, 192) of the head = e4m3(hi), [192, 256) = e4m3(lo * 2^12): the operands of the two correction products on the
    // block-scaled fp8 MFMA (f3r_attn_asm_qk3f8_f16).  v_cvt_pk_fp8_f32 does not saturate (out of range -> NaN): clamp to +-448 first.
    float h4[4] = {lo_f<T>(hi[0]), hi_f<T>(hi[0]), lo_f<T>(hi[1]), hi_f<T>(hi[1])};
    float l4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      l4[i] = fminf(fmaxf((v[i] - h4[i]) * 4096.0f, -448.f), 448.f);
      h4[i] = fminf(fmaxf(h4[i], -448.f), 448.f);
    }
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(h4[0], h4[1], 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(h4[2], h4[3], w, true);
    int wl = __builtin_amdgcn_cvt_pk_fp8_f32(l4[0], l4[1], 0, false);
    wl = __builtin_amdgcn_cvt_pk_fp8_f32(l4[2], l4[3], wl, true);
    uint8_t* ob = (uint8_t*)(out + (r * n_heads + h) * 128) + 128 + d;   // d is a multiple of 4
    *(int*)ob = w;
    *(int*)(ob + 64) = wl;
  }
}

// V fp32 [n_seq * tk][ld] (columns 0 .. n of `in`) -> V^T [n_seq][n][ldvt], one plane; 64 x 64 tiles through LDS
template <class T>
__global__ __launch_bounds__(256) void transpose_rows_kernel(const float* __restrict__ in, int64_t ld, int64_t tk, int n, int64_t ldvt, uint16_t* __restrict__ out) {
  __shared__ float tile[64][65];
  const int64_t seq = blockIdx.z;
  const int64_t k0 = (int64_t)blockIdx.x * 64;
  const int d0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int64_t key = k0 + i;
    tile[i][tx] = key < tk ? in[(seq * tk + key) * ld + d0 + tx] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) out[(seq * n + d0 + i) * ldvt + k0 + tx] = to_lp<T>(tile[tx][i]);
}

// one thread per 4 consecutive columns of a (row, head)
template <class T>
__global__ void state_finish_kernel(const float* __restrict__ st_o, const float* __restrict__ st_ml, int64_t rows, int n_heads, int head_dim,
                                    uint16_t* __restrict__ o_hi, uint16_t* __restrict__ o_lo, float* __restrict__ o_f32, int64_t ldo) {
  const int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int64_t n = (int64_t)n_heads * head_dim;
  if (idx >= rows * n) return;
  const int64_t r = idx / n;
  const int c = (int)(idx - r * n);
  const int h = c / head_dim;
  const float* ml = st_ml + (r * n_heads + h) * 4;
  const float inv = 1.0f / (ml[1] + ml[2]);
  float4v v = *(const float4v*)(st_o + idx);
  v *= inv;
  if (o_f32) *(float4v*)(o_f32 + r * ldo + c) = v;
  if (o_hi) {
    u32x2 hi;
    hi[0] = pack2<T>(v[0], v[1]);
    hi[1] = pack2<T>(v[2], v[3]);
    *(u32x2*)(o_hi + r * ldo + c) = hi;
    if (o_lo) {
      u32x2 lo;
      lo[0] = pack2<T>(v[0] - lo_f<T>(hi[0]), v[1] - hi_f<T>(hi[0]));
      lo[1] = pack2<T>(v[2] - lo_f<T>(hi[1]), v[3] - hi_f<T>(hi[1]));
      *(u32x2*)(o_lo + r * ldo + c) = lo;
    }
  }
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
inline bool al8(const void* p) { return (((uintptr_t)p) & 7) == 0; }

}  // namespace

extern "C" int f3r_qkv_planes(const float* qkv, int64_t ld, int64_t n_seq, int64_t seq_len, int n_heads, int kv_heads, float q_scale, void* q_planes,
                              void* k_planes, void* vt, int64_t ldvt, int dtype, int planes, f3r_stream_t stream) {
  F3R_REQUIRE(planes == 2 || (planes == 3 && dtype == F3R_F16), "f3r_qkv_planes: planes %d (2 = [hi | lo] 16-bit; 3 = [hi fp16 | e4m3(hi) | e4m3(lo 2^12)], fp16 only)", planes);
  F3R_REQUIRE(qkv && q_planes && k_planes && vt, "f3r_qkv_planes: null pointer");
  F3R_REQUIRE(n_seq > 0 && seq_len >= 0 && n_heads > 0 && kv_heads > 0 && n_heads % kv_heads == 0, "f3r_qkv_planes: bad sizes");
  F3R_REQUIRE(dtype == F3R_F16 || dtype == F3R_BF16, "f3r_qkv_planes: bad dtype %d", dtype);
  const int64_t Dq = (int64_t)n_heads * 64, Dk = (int64_t)kv_heads * 64;
  F3R_REQUIRE(al16(qkv) && ld % 4 == 0 && ld >= Dq + 2 * Dk && al8(q_planes) && al8(k_planes), "f3r_qkv_planes: alignment / row stride");
  F3R_REQUIRE(ldvt >= seq_len && ldvt % 64 == 0, "f3r_qkv_planes: ldvt %lld must be a multiple of 64 and >= seq_len", (long long)ldvt);
  const int64_t rows = n_seq * seq_len;
  if (rows == 0) return F3R_OK;
  F3R_REQUIRE(rows * Dq / 4 / 256 < (1ll << 31) && ldvt / 64 < (1ll << 31) && n_seq < 65536 && kv_heads < 65536, "f3r_qkv_planes: grid too large");
  hipStream_t s = (hipStream_t)stream;
  const dim3 gq((unsigned)((rows * Dq / 4 + 255) / 256)), gk((unsigned)((rows * Dk / 4 + 255) / 256));
  const dim3 gv((unsigned)(ldvt / 64), (unsigned)kv_heads, (unsigned)n_seq);
#define F3R_QP(TT, F8)                                                                                                                          \
  hipLaunchKernelGGL((planes_rows_kernel<TT, F8>), gq, dim3(256), 0, s, qkv, ld, rows, n_heads, q_scale, (uint16_t*)q_planes);                  \
  hipLaunchKernelGGL((planes_rows_kernel<TT, F8>), gk, dim3(256), 0, s, qkv + Dq, ld, rows, kv_heads, 1.0f, (uint16_t*)k_planes);               \
  hipLaunchKernelGGL(transpose_rows_kernel<TT>, gv, dim3(256), 0, s, qkv + Dq + Dk, ld, seq_len, (int)Dk, ldvt, (uint16_t*)vt)
  if (planes == 3) { F3R_QP(F16, true); } else if (dtype == F3R_F16) { F3R_QP(F16, false); } else { F3R_QP(BF16, false); }
#undef F3R_QP
  return f3r_check_launch("f3r_qkv_planes");
}

extern "C" int f3r_attn_state_finish(const float* st_o, const float* st_ml, int64_t rows, int n_heads, int head_dim, void* o_hi, void* o_lo, float* o_f32,
                                     int64_t ldo, int dtype, f3r_stream_t stream) {
  F3R_REQUIRE(st_o && st_ml && (o_hi || o_f32), "f3r_attn_state_finish: null pointer");
  F3R_REQUIRE(!o_lo || o_hi, "f3r_attn_state_finish: a low plane needs its high plane");
  F3R_REQUIRE(rows >= 0 && n_heads > 0 && head_dim > 0 && head_dim % 4 == 0, "f3r_attn_state_finish: bad sizes");
  F3R_REQUIRE(dtype == F3R_F16 || dtype == F3R_BF16, "f3r_attn_state_finish: bad dtype %d", dtype);
  const int64_t n = (int64_t)n_heads * head_dim;
  F3R_REQUIRE(al16(st_o) && al16(st_ml) && ldo >= n && ldo % 4 == 0 && al8(o_hi) && al8(o_lo) && al16(o_f32), "f3r_attn_state_finish: alignment / row stride");
  if (rows == 0) return F3R_OK;
  F3R_REQUIRE(rows * n / 4 / 256 < (1ll << 31), "f3r_attn_state_finish: grid too large");
  const dim3 g((unsigned)((rows * n / 4 + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == F3R_F16)
    hipLaunchKernelGGL(state_finish_kernel<F16>, g, dim3(256), 0, s, st_o, st_ml, rows, n_heads, head_dim, (uint16_t*)o_hi, (uint16_t*)o_lo, o_f32, ldo);
  else
    hipLaunchKernelGGL(state_finish_kernel<BF16>, g, dim3(256), 0, s, st_o, st_ml, rows, n_heads, head_dim, (uint16_t*)o_hi, (uint16_t*)o_lo, o_f32, ldo);
  return f3r_check_launch("f3r_attn_state_finish");
}
