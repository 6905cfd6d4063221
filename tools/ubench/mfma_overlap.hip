// Does the gfx950 matrix pipe overlap with VALU work (a) across two waves of one SIMD, (b) inside one wave?
// One 512-thread workgroup per CU: waves 0-3 ("slot 0", one per SIMD) and waves 4-7 ("slot 1") each run a role:
//   0 idle   1 MFMA stream (32x32x16 bf16, 4 independent accumulators)   2 v_fma_f32 stream   3 v_exp_f32 stream
//   4 one MFMA followed by F independent v_fma_f32 (F = filler count), repeated   5 ds_read_b128 -> wait -> 2 MFMA
//   6 like 5 with the reads issued two steps ahead
// Prints s_memtime ticks per loop step for wave 0 (slot 0) and wave 4 (slot 1) of workgroup 0.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_overlap.hip -o mfma_overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float float16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define STEPS 4096

template <int F>
__device__ __forceinline__ void fillers(float (&a)[8]) {
#pragma unroll
  for (int i = 0; i < F; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i & 7]));
}

template <int ROLE, int F>
__device__ void run_role(float* out, uint64_t* cyc, int slot, const uint32_t* lds) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = 1e-3f * (threadIdx.x + i);
  float16v acc[4];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  u32x4 fa = {threadIdx.x, 1u, 2u, 3u}, fb = {7u, threadIdx.x, 5u, 4u};
  const int lane = threadIdx.x & 63;
  const uint64_t t0 = __builtin_readcyclecounter();
  if (ROLE == 1) {
    for (int s = 0; s < STEPS / 4; ++s) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa), __builtin_bit_cast(bf8, fb), acc[j], 0, 0, 0);
    }
  } else if (ROLE == 2) {
    for (int s = 0; s < STEPS / 8; ++s) fillers<8>(a);
  } else if (ROLE == 3) {
    for (int s = 0; s < STEPS / 8; ++s) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
    }
  } else if (ROLE == 4) {
    for (int s = 0; s < STEPS / 4; ++s) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa), __builtin_bit_cast(bf8, fb), acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        fillers<F>(a);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if (ROLE == 7) {
    for (int s = 0; s < STEPS / 4; ++s) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa), __builtin_bit_cast(bf8, fb), acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < F; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i & 7]));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if (ROLE == 8) {  // softmax-like mix per MFMA: F x (2 exp + 1 cvt_pk + 1 dot2)
    for (int s = 0; s < STEPS / 4; ++s) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa), __builtin_bit_cast(bf8, fb), acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < F; ++i) {
          asm volatile("v_exp_f32 %0, %0" : "+v"(a[(2 * i) & 7]));
          asm volatile("v_exp_f32 %0, %0" : "+v"(a[(2 * i + 1) & 7]));
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(fa[i & 3]) : "v"(a[(2 * i) & 7]), "v"(a[(2 * i + 1) & 7]));
          asm volatile("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(a[(i + 5) & 7]) : "v"(fa[i & 3]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if (ROLE == 9) {  // the xp slot mix with independent operands: MFMA, cvt_pk, exp, exp, dot2 (x F)
    float d0 = 0.f, d1 = 0.f;
    for (int s = 0; s < STEPS / 4; ++s) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa), __builtin_bit_cast(bf8, fb), acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < F; ++i) {
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(fb[(j + i) & 1 ? 2 : 3]) : "v"(a[4]), "v"(a[5]));
          asm volatile("v_exp_f32 %0, %1" : "=v"(a[(2 * j) & 3]) : "v"(a[6]));
          asm volatile("v_exp_f32 %0, %1" : "=v"(a[(2 * j + 1) & 3]) : "v"(a[7]));
          asm volatile("v_dot2c_f32_bf16 %0, %1, %1" : "+v"((j + i) & 1 ? d0 : d1) : "v"(fa[1]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    a[0] += d0 + d1;
  } else if (ROLE == 10) {  // single-instruction streams: F = 0 cvt_pk_bf16, 1 dot2c, 2 v_xor, 3 v_lshl_or, 4 ds_read_b128 (no wait)
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < STEPS / 8; ++s) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (F == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(fb[i & 3]) : "v"(a[i & 7]), "v"(a[(i + 1) & 7]));
        if (F == 1) asm volatile("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(d[i & 3]) : "v"(fa[i & 3]));
        if (F == 2) asm volatile("v_xor_b32 %0, 48, %1" : "=v"(fb[i & 3]) : "v"(fa[i & 3]));
        if (F == 3) asm volatile("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(fb[i & 3]) : "v"(fa[i & 3]), "s"(s));
      }
    }
    a[0] += d[0] + d[1] + d[2] + d[3];
  } else if (ROLE == 5) {
    for (int s = 0; s < STEPS / 2; ++s) {
      const u32x4 r = *(const volatile u32x4*)(lds + ((lane * 4 + s * 64) & 4095));
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, r), __builtin_bit_cast(bf8, fb), acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, r), __builtin_bit_cast(bf8, fa), acc[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (ROLE == 6) {
    u32x4 r0 = *(const volatile u32x4*)(lds + (lane * 4 & 4095));
    u32x4 r1 = *(const volatile u32x4*)(lds + ((lane * 4 + 64) & 4095));
    for (int s = 0; s < STEPS / 2; s += 2) {
      const u32x4 n0 = *(const volatile u32x4*)(lds + ((lane * 4 + (s + 2) * 64) & 4095));
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, r0), __builtin_bit_cast(bf8, fb), acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, r0), __builtin_bit_cast(bf8, fa), acc[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const u32x4 n1 = *(const volatile u32x4*)(lds + ((lane * 4 + (s + 3) * 64) & 4095));
      __builtin_amdgcn_sched_barrier(0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, r1), __builtin_bit_cast(bf8, fb), acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, r1), __builtin_bit_cast(bf8, fa), acc[3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      r0 = n0;
      r1 = n1;
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float sum = 0.f;
  for (int i = 0; i < 8; ++i) sum += a[i];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 16; ++i) sum += acc[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (blockIdx.x == 0 && lane == 0 && (threadIdx.x >> 6) == slot * 4) cyc[slot] = t1 - t0;
}

template <int R0, int R1, int F>
__global__ __launch_bounds__(512) void k(float* out, uint64_t* cyc) {
  __shared__ uint32_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i * 2654435761u;
  __syncthreads();
  const int slot = threadIdx.x >> 8;
  if (slot == 0) {
    if (R0) run_role<R0, F>(out, cyc, 0, lds);
  } else {
    if (R1) run_role<R1, F>(out, cyc, 1, lds);
  }
}

template <int R0, int R1, int F>
void go(const char* what, float* out, uint64_t* cyc, int units0, int units1) {
  hipMemset(cyc, 0, 16);
  hipLaunchKernelGGL((k<R0, R1, F>), dim3(256), dim3(512), 0, 0, out, cyc);
  hipDeviceSynchronize();
  uint64_t c[2];
  hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
  printf("%-58s slot0 %7.2f ticks/step   slot1 %7.2f ticks/step\n", what, (double)c[0] / units0, (double)c[1] / units1);
}

int main() {
  float* out;
  uint64_t* cyc;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 16);
  const int S = STEPS;
  go<1, 0, 0>("MFMA stream alone (per MFMA)", out, cyc, S, 1);
  go<2, 0, 0>("v_fma stream alone (per instr)", out, cyc, S, 1);
  go<3, 0, 0>("v_exp stream alone (per instr)", out, cyc, S, 1);
  go<1, 1, 0>("MFMA || MFMA (two waves of a SIMD)", out, cyc, S, S);
  go<1, 2, 0>("MFMA (slot0) || v_fma (slot1)", out, cyc, S, S);
  go<1, 3, 0>("MFMA (slot0) || v_exp (slot1)", out, cyc, S, S);
  go<2, 2, 0>("v_fma || v_fma", out, cyc, S, S);
  go<3, 3, 0>("v_exp || v_exp", out, cyc, S, S);
  go<3, 2, 0>("v_exp || v_fma", out, cyc, S, S);
  go<4, 0, 0>("one wave: MFMA + 0 fillers (per group)", out, cyc, S, 1);
  go<4, 0, 2>("one wave: MFMA + 2 v_fma", out, cyc, S, 1);
  go<4, 0, 4>("one wave: MFMA + 4 v_fma", out, cyc, S, 1);
  go<4, 0, 6>("one wave: MFMA + 6 v_fma", out, cyc, S, 1);
  go<4, 0, 8>("one wave: MFMA + 8 v_fma", out, cyc, S, 1);
  go<4, 0, 12>("one wave: MFMA + 12 v_fma", out, cyc, S, 1);
  go<4, 4, 4>("two waves, each MFMA + 4 v_fma", out, cyc, S, S);
  go<4, 4, 8>("two waves, each MFMA + 8 v_fma", out, cyc, S, S);
  go<4, 4, 12>("two waves, each MFMA + 12 v_fma", out, cyc, S, S);
  go<3, 1, 0>("v_exp (slot0, older) || MFMA (slot1)", out, cyc, S, S);
  go<2, 1, 0>("v_fma (slot0, older) || MFMA (slot1)", out, cyc, S, S);
  go<7, 0, 1>("one wave: MFMA + 1 v_exp", out, cyc, S, 1);
  go<7, 0, 2>("one wave: MFMA + 2 v_exp", out, cyc, S, 1);
  go<7, 0, 3>("one wave: MFMA + 3 v_exp", out, cyc, S, 1);
  go<7, 0, 4>("one wave: MFMA + 4 v_exp", out, cyc, S, 1);
  go<8, 0, 1>("one wave: MFMA + 1 x (2 exp, cvt_pk, dot2)", out, cyc, S, 1);
  go<8, 0, 2>("one wave: MFMA + 2 x (2 exp, cvt_pk, dot2)", out, cyc, S, 1);
  go<8, 8, 1>("two waves: MFMA + 1 x (2 exp, cvt_pk, dot2)", out, cyc, S, S);
  go<8, 8, 2>("two waves: MFMA + 2 x (2 exp, cvt_pk, dot2)", out, cyc, S, S);
  go<7, 7, 2>("two waves: MFMA + 2 v_exp", out, cyc, S, S);
  go<4, 7, 2>("MFMA + 2 v_fma (slot0) || MFMA + 2 v_exp (slot1)", out, cyc, S, S);
  go<10, 0, 0>("v_cvt_pk_bf16_f32 stream alone", out, cyc, S, 1);
  go<10, 0, 1>("v_dot2c_f32_bf16 stream alone", out, cyc, S, 1);
  go<10, 0, 2>("v_xor_b32 stream alone", out, cyc, S, 1);
  go<10, 0, 3>("v_lshl_or_b32 stream alone", out, cyc, S, 1);
  go<9, 0, 1>("one wave: MFMA + (cvt, exp, exp, dot2) independent", out, cyc, S, 1);
  go<9, 9, 1>("two waves: MFMA + (cvt, exp, exp, dot2) independent", out, cyc, S, S);
  go<9, 0, 2>("one wave: MFMA + 2 x (cvt, exp, exp, dot2) independent", out, cyc, S, 1);
  go<5, 0, 0>("one wave: ds_read -> wait -> 2 MFMA (per 2-MFMA step)", out, cyc, S / 2, 1);
  go<6, 0, 0>("one wave: same, reads 2 steps ahead", out, cyc, S / 2, 1);
  go<5, 5, 0>("two waves: ds_read -> wait -> 2 MFMA", out, cyc, S / 2, S / 2);
  go<6, 6, 0>("two waves: same, reads 2 steps ahead", out, cyc, S / 2, S / 2);
  go<5, 2, 0>("ds_read->2 MFMA (slot0) || v_fma (slot1)", out, cyc, S / 2, S);
  return 0;
}
