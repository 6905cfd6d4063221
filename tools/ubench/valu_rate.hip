// VALU / transcendental issue-rate micro-benchmark for gfx950: cycles per wave-instruction at 1, 2, 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 64
#define OUTER 256

template <int OP>
__global__ void k(float* out, uint64_t* cyc, float seed) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 b[4];
  for (int i = 0; i < 4; ++i) b[i] = f2{a[2 * i], a[2 * i + 1]};
  uint64_t t0 = __builtin_readcyclecounter();
  for (int o = 0; o < OUTER; ++o) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
        if (OP == 2) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[i]));
        if (OP == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
        if (OP == 4) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
        if (OP == 5) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
        if (OP == 6) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(b[i & 3]));
        if (OP == 7) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(b[i & 3]));
        if (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 9) asm volatile("v_ldexp_f32 %0, %0, 1" : "+v"(a[i]));
        if (OP == 10) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
        if (OP == 11) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
        if (OP == 12) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(b[i & 3]));
        if (OP == 13) asm volatile("v_sub_f32 %0, %0, %0" : "+v"(a[i]));
      }
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  float acc = 0;
  for (int i = 0; i < 8; ++i) acc += a[i];
  for (int i = 0; i < 4; ++i) acc += b[i][0] + b[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// mixed: exp + fma interleaved 1:1 (does the transcendental unit overlap the main VALU?)
__global__ void kmix(float* out, uint64_t* cyc, float seed) {
  float a[8], c[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; c[i] = a[i] * 0.5f; }
  uint64_t t0 = __builtin_readcyclecounter();
  for (int o = 0; o < OUTER; ++o) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(c[i]));
      }
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  float acc = 0;
  for (int i = 0; i < 8; ++i) acc += a[i] + c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
void run(const char* name, float* out, uint64_t* cyc) {
  for (int waves_per_simd : {1, 2, 4}) {
    int threads = 64 * 4 * waves_per_simd;  // one block per CU fills `waves_per_simd` waves on each of the 4 SIMDs
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 0.001f);
    hipDeviceSynchronize();
    uint64_t c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double per = (double)c / (OUTER * REP);
    printf("%-18s waves/SIMD=%d  cycles(s_memtime ticks)/wave-instr = %6.2f   => per-SIMD issue interval = %5.2f\n", name, waves_per_simd, per, per / waves_per_simd);
  }
}

int main() {
  float* out; uint64_t* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  run<0>("v_exp_f32", out, cyc); run<4>("v_exp_f16", out, cyc); run<8>("v_rcp_f32", out, cyc);
  run<1>("v_fma_f32", out, cyc); run<5>("v_add_f32", out, cyc); run<13>("v_sub_f32", out, cyc); run<2>("v_max3_f32", out, cyc);
  run<3>("v_cvt_pk_bf16_f32", out, cyc); run<6>("v_pk_fma_f32", out, cyc); run<7>("v_pk_add_f32", out, cyc); run<12>("v_pk_mul_f32", out, cyc);
  run<9>("v_ldexp_f32", out, cyc); run<10>("v_fract_f32", out, cyc); run<11>("v_cvt_i32_f32", out, cyc);
  for (int w : {1, 2, 4}) {
    hipLaunchKernelGGL(kmix, dim3(256), dim3(64 * 4 * w), 0, 0, out, cyc, 0.001f);
    hipDeviceSynchronize();
    uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("exp+fma pair       waves/SIMD=%d  cycles per PAIR = %6.2f\n", w, (double)c / (OUTER * REP));
  }
  return 0;
}
