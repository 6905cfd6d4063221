// Go / no-go measurement for "cheaper correction planes" (VERDICT r04 item 1c): what does the gfx950 matrix pipe sustain,
// under the package power cap and on random data, when the fp16 MFMAs of a split-precision GEMM (W2: A W_hi + A W_lo, X3: + A_lo W_hi) are
// mixed with block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 instructions that would carry the low planes in fp8 / fp6 / fp4?
//
// One 256-thread workgroup per CU (160 KiB of dynamic LDS keeps it alone there, one wave per SIMD like the hand-scheduled kernels); a wave
// owns 4 x 4 blocks of 32 x 32 accumulators (256 registers) and runs, per "K-tile" (64 deep) and block,
//      NF16 x v_mfma_f32_32x32x16_f16   (4 = one fp16 plane of the K-tile)      +      NF8 x v_mfma_scale_f32_32x32x64_<fmt> (1 = one plane),
// k-step major (16 blocks per k-step, as gemm_gen.py issues them), on operand fragments that differ per k-step (random fp16 / random
// fp8-fp6-fp4 bit patterns without NaN codes).  No LDS / global traffic inside the loop: this is the ceiling of the pipe + power cap, which
// is what decides the question.  Reported per mode: wall time, executed TFLOP/s, the rate in "K-tile blocks per second", the shader clock
// from s_memtime / s_memrealtime of wave 0, and matrix-pipe cycles per K-tile block.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_mixed.hip -o tools/ubench/mfma_mixed ; run on the GPU box (prints JSON lines).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

// FMT: 0 = fp8 e4m3, 1 = bf8 e5m2, 2 = fp6 e2m3, 3 = bf6 e3m2, 4 = fp4 e2m1 (cbsz / blgp of the instruction)
template <int NF16, int NF8, int FMT>
__global__ __launch_bounds__(256) void mixed_kernel(const f16x8* __restrict__ h16, const i32x8* __restrict__ h8, float* __restrict__ out, uint64_t* __restrict__ clk,
                                                    int iters) {
  extern __shared__ char lds_hold[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[16];
#pragma unroll
  for (int b = 0; b < 16; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
  // fragments: fp16 [operand][block][k-step], low-precision [operand][block]
  f16x8 a16[4][4], b16[4][4];
  i32x8 a8[4], b8[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      a16[i][ks] = h16[((wave * 2 + 0) * 16 + i * 4 + ks) * 64 + lane];
      b16[i][ks] = h16[((wave * 2 + 1) * 16 + i * 4 + ks) * 64 + lane];
    }
    a8[i] = h8[((wave * 2 + 0) * 4 + i) * 64 + lane];
    b8[i] = h8[((wave * 2 + 1) * 4 + i) * 64 + lane];
  }
  const int one = 0x7F7F7F7F;  // E8M0 scale 1.0 in every byte
  __builtin_amdgcn_s_barrier();
  const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < NF16 / 4; ++rep) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ib * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[ib][ks], b16[j][ks], acc[ib * 4 + j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int rep = 0; rep < NF8; ++rep) {
#pragma unroll
      for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[ib * 4 + j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[ib], b8[j], acc[ib * 4 + j], FMT, FMT, 0, one, 0, one);
    }
  }
  const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < 16; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[b][i];
  out[(blockIdx.x * 256 + threadIdx.x)] = s + (lds_hold[0] == 77 ? 1.f : 0.f);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = r1 - r0;
  }
}

typedef void (*kern_t)(const f16x8*, const i32x8*, float*, uint64_t*, int);

struct Mode {
  const char* name;
  kern_t fn;
  int nf16, nf8;
  const char* fmt;
  const char* what;
};

#define MODE(nf16, nf8, fmt, fname, what) \
  { "f16x" #nf16 "+" fname "x" #nf8, mixed_kernel<nf16, nf8, fmt>, nf16, nf8, fname, what }

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 1.0;  // measured time per mode (after a warm-up of the same length / 3)
  int dev = 0, cus = 0, wall_khz = 0;
  CHECK(hipGetDevice(&dev));
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  CHECK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev));
  std::vector<Mode> modes = {
      MODE(4, 0, 0, "none", "one fp16 plane (precision fast)"),
      MODE(8, 0, 0, "none", "W2 today: two fp16 planes"),
      MODE(12, 0, 0, "none", "X3 today: three fp16 products"),
      MODE(4, 1, 0, "fp8", "W2 with the low plane in fp8 e4m3 (1 : 1 per K-tile block = 4 : 1 instructions)"),
      MODE(4, 1, 2, "fp6", "W2 with the low plane in fp6 e2m3"),
      MODE(4, 1, 4, "fp4", "W2 with the low plane in fp4 e2m1"),
      MODE(4, 2, 0, "fp8", "X3 with both low products in fp8 e4m3 (2 : 1 of the planes)"),
      MODE(4, 2, 2, "fp6", "X3 with both low products in fp6 e2m3"),
      MODE(0, 1, 0, "fp8", "fp8 e4m3 alone"),
      MODE(0, 1, 2, "fp6", "fp6 e2m3 alone"),
      MODE(0, 1, 4, "fp4", "fp4 e2m1 alone"),
  };
  // operands: random fp16 in N(0,1)-like range (uniform +-2), random low-precision bit patterns with the fp8 NaN codes (0x7F / 0xFF) avoided
  const size_t n16 = 4 * 2 * 16 * 64, n8 = 4 * 2 * 4 * 64;
  std::vector<f16x8> h16(n16);
  std::vector<i32x8> h8(n8);
  srand(1234);
  for (auto& v : h16)
    for (int i = 0; i < 8; ++i) v[i] = (_Float16)(4.0f * rand() / RAND_MAX - 2.0f);
  for (auto& v : h8)
    for (int i = 0; i < 8; ++i) {
      uint32_t w = 0;
      for (int b = 0; b < 4; ++b) {
        uint32_t x = rand() & 0xFF;
        if ((x & 0x7F) == 0x7F) x ^= 1;
        w |= x << (8 * b);
      }
      v[i] = (int)w;
    }
  f16x8* d16;
  i32x8* d8;
  float* dout;
  uint64_t* dclk;
  CHECK(hipMalloc(&d16, n16 * sizeof(f16x8)));
  CHECK(hipMalloc(&d8, n8 * sizeof(i32x8)));
  CHECK(hipMalloc(&dout, (size_t)cus * 256 * sizeof(float)));
  CHECK(hipMalloc(&dclk, 2 * sizeof(uint64_t)));
  CHECK(hipMemcpy(d16, h16.data(), n16 * sizeof(f16x8), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d8, h8.data(), n8 * sizeof(i32x8), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int lds = 160 * 1024;
  double base_w2 = 0, base_x3 = 0;
  for (auto& m : modes) {
    CHECK(hipFuncSetAttribute((const void*)m.fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    // cycles of the matrix pipe per iteration (16 blocks): 32 per fp16 32x32x16; 64 (fp8) or 32 (fp6 / fp4) per scaled 32x32x64 at the dense peaks
    const double flop_it = 4.0 * 16 * (m.nf16 * 2.0 * 32 * 32 * 16 + m.nf8 * 2.0 * 32 * 32 * 64);  // per workgroup and iteration
    int iters = 2000;
    // calibrate the iteration count to the requested duration, then warm up and measure (the power cap settles within the warm-up)
    for (int pass = 0; pass < 3; ++pass) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(m.fn, dim3(cus), dim3(256), lds, 0, d16, d8, dout, dclk, iters);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (pass == 0) iters = (int)(iters * (seconds / 3 * 1e3) / ms) + 1;
      if (pass == 1) iters = (int)(iters * 3.0) + 1;
      if (pass == 2) {
        uint64_t clk[2];
        CHECK(hipMemcpy(clk, dclk, sizeof(clk), hipMemcpyDeviceToHost));
        const double tf = flop_it * iters * cus / (ms * 1e-3) / 1e12;
        const double blocks_per_s = 16.0 * 4 * cus * iters / (ms * 1e-3);  // K-tile blocks (32 x 32 outputs, 64 deep, all planes) per second
        const double ghz = (double)clk[0] / ((double)clk[1] / (wall_khz * 1e3)) / 1e9;
        const double cyc_per_block = (double)clk[0] / ((double)iters * 16);
        if (m.nf16 == 8 && m.nf8 == 0) base_w2 = blocks_per_s;
        if (m.nf16 == 12 && m.nf8 == 0) base_x3 = blocks_per_s;
        const double base = m.nf16 == 4 && m.nf8 == 1 ? base_w2 : (m.nf16 == 4 && m.nf8 == 2 ? base_x3 : 0);
        printf("{\"mode\": \"%s\", \"what\": \"%s\", \"ms\": %.2f, \"iters\": %d, \"tflops_executed\": %.1f, \"ktile_blocks_per_s\": %.4g, "
               "\"shader_clock_ghz\": %.3f, \"cycles_per_ktile_block\": %.1f, \"speedup_vs_all_fp16_planes\": %s%.3f, \"wall_clock_khz\": %d, \"cus\": %d}\n",
               m.name, m.what, ms, iters, tf, blocks_per_s, ghz, cyc_per_block, base > 0 ? "" : "0", base > 0 ? blocks_per_s / base : 0.0, wall_khz, cus);
        fflush(stdout);
      }
    }
  }
  return 0;
}
