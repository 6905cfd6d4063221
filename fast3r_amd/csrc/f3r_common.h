// Shared device helpers for the gfx950 (CDNA4) kernels.  wave = 64 lanes; MFMA operands are 16-bit
// (f16 or bf16), accumulation is fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/f3r.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

struct F16 {
  typedef _Float16 elem;
  typedef half8 vec8;
  static constexpr int id = F3R_F16;
  static __device__ __forceinline__ float4v mfma16(vec8 a, vec8 b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float16v mfma32(vec8 a, vec8 b, float16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  // 32 x 32 x 8 MFMA (2-register operands; lane (row, g = l >> 5) supplies k = 4 g .. 4 g + 3): used for a 2-slot bias step
  static __device__ __forceinline__ float16v mfma32k8(u32x2 a, u32x2 b, float16v c) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    return __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_bit_cast(h4, a), __builtin_bit_cast(h4, b), c, 0, 0, 0);
  }
  // acc + lo(pk) + hi(pk): one v_dot2c_f32_f16 against (1, 1)
  static __device__ __forceinline__ float sum2(uint32_t pk, float acc) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 ones = {(_Float16)1.0f, (_Float16)1.0f};
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, pk), ones, acc, false);
  }
};
struct BF16 {
  typedef __bf16 elem;
  typedef bf8 vec8;
  static constexpr int id = F3R_BF16;
  static __device__ __forceinline__ float4v mfma16(vec8 a, vec8 b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float16v mfma32(vec8 a, vec8 b, float16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float16v mfma32k8(u32x2 a, u32x2 b, float16v c) {
    typedef short s4 __attribute__((ext_vector_type(4)));
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s4, a), __builtin_bit_cast(s4, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float sum2(uint32_t pk, float acc) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 ones = {(__bf16)1.0f, (__bf16)1.0f};
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, pk), ones, acc, false);
  }
};

// fp32 -> lowp, round to nearest even (v_cvt_f16_f32 / v_cvt_pk_bf16_f32 on gfx950)
template <class T>
__device__ __forceinline__ uint16_t to_lp(float x) {
  typename T::elem e = (typename T::elem)x;
  return __builtin_bit_cast(uint16_t, e);
}
template <class T>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef typename T::elem e2 __attribute__((ext_vector_type(2)));
  e2 p;
  p[0] = (typename T::elem)lo;
  p[1] = (typename T::elem)hi;
  return __builtin_bit_cast(uint32_t, p);
}
template <class T>
__device__ __forceinline__ float from_lp(uint16_t b) {
  return (float)__builtin_bit_cast(typename T::elem, b);
}
template <class T>
__device__ __forceinline__ float lo_f(uint32_t w) { return from_lp<T>((uint16_t)(w & 0xffffu)); }
template <class T>
__device__ __forceinline__ float hi_f(uint32_t w) { return from_lp<T>((uint16_t)(w >> 16)); }

// ReLU on two packed 16-bit floats of either type: clear a half whose sign bit is set.
__device__ __forceinline__ uint32_t relu_pk(uint32_t w) {
  uint32_t neg = (w >> 15) & 0x00010001u;
  return w & ~(neg * 0xffffu);
}

template <class T>
__device__ __forceinline__ typename T::vec8 as_vec8(u32x4 v) {
  return __builtin_bit_cast(typename T::vec8, v);
}

// f3r_gemm256.hip: the 256x256-tile kernel behind f3r_gemm for large regular shapes
bool f3r_gemm256_eligible(const f3r_gemm_args& a);   // can take the problem at all
bool f3r_gemm256_preferred(const f3r_gemm_args& a);  // ... and is expected to be faster than the 128-tile kernel
int f3r_gemm256_launch(const f3r_gemm_args& a, hipStream_t stream, int stagger);
int f3r_gemm256_lab(const f3r_gemm_args& a, hipStream_t stream);  // -DF3R_GEMM_LAB builds only (tools/lab)

// f3r_gemm_asm.hip: the hand-scheduled GEMM kernels (csrc/asm/gemm_gen.py) behind f3r_gemm
bool f3r_gemm_asm_eligible(const f3r_gemm_args& a, const char** why);
bool f3r_gemm_asm_preferred(int64_t tiles);  // does a launch of that many 256 x 256 tiles fill the persistent grid well enough?
int f3r_gemm_asm_launch(const f3r_gemm_args& a, hipStream_t stream);
bool f3r_gemm_asm_f8_eligible(const f3r_gemm_args& a, const char** why);   // F3R_SPLIT_W2F8: the kernels with the low plane in fp8
int f3r_gemm_asm_f8_launch(const f3r_gemm_args& a, hipStream_t stream);
bool f3r_gemm_asm_qkv_eligible(const f3r_gemm_args& a, const char** why);  // the QKV projection without rotary embedding, as two launches
int f3r_gemm_asm_qkv_launch(const f3r_gemm_args& a, hipStream_t stream);

// f3r_attn_asm.hip: the hand-scheduled attention kernel (csrc/asm/attn_gen.py) behind f3r_attn_fwd
bool f3r_attn_asm_eligible(const f3r_attn_args& a, int64_t min_keys, const char** why);
int f3r_attn_asm_launch(const f3r_attn_args& a, hipStream_t stream);
bool f3r_attn_asm_uses_q256(const f3r_attn_args& a);   // the 256-query form of the head_dim-64 kernel (small launches)
bool f3r_attn_asm_splits_tail(const f3r_attn_args& a); // two launches: whole rounds of 512-query items + the last, partly filled round as 256-query items

// f3r_attn_generic.hip: head_dim != 64
int f3r_attn_generic_launch(const f3r_attn_args& a, hipStream_t stream);

// host-side error plumbing (f3r_capi.cpp)
void f3r_set_error(const char* fmt, ...);
int f3r_check_launch(const char* what);

#define F3R_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      f3r_set_error(__VA_ARGS__);     \
      return F3R_ERR_ARG;             \
    }                                 \
  } while (0)
