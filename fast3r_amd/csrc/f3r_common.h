// Shared device helpers for the gfx950 (CDNA4) kernels.  wave = 64 lanes; MFMA operands are 16-bit
// (f16 or bf16), accumulation is fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/f3r.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

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
