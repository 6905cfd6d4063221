// Hardware probe for the block-scaled fp8 MFMA used by the split-precision GEMM's low plane (csrc/asm/gemm_gen.py, role w2f8):
// one wave runs v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3) on random lane registers and dumps operands and result, so that the
// lane -> (row, k) layout and the scale-byte semantics can be read off on the host (tools/ubench/mfma_scale_probe.py tests the candidate
// layouts against a float64 product of the decoded bytes); plus v_cvt_pk_fp8_f32 on a list of floats (rounding, saturation, word select).
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_scale_probe.hip -o tools/ubench/mfma_scale_probe ; prints JSON lines.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// mode 0: unit scales; 1: per-lane scale_a (byte 0 of each lane's register differs), opsel 0; 2: per-lane scale_b; 3: scale_a bytes differ, opsel 1
template <int OPSEL_A, int OPSEL_B>
__global__ void probe(const int* a_regs, const int* b_regs, const int* sa, const int* sb, float* d) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = a_regs[lane * 8 + i];
    b[i] = b_regs[lane * 8 + i];
  }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPSEL_A, sa[lane], OPSEL_B, sb[lane]);
  for (int i = 0; i < 16; ++i) d[lane * 16 + i] = c[i];
}

__global__ void cvt_probe(const float* x, int n, int* out) {
  const int i = threadIdx.x;
  if (i * 2 + 1 < n) {
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0x55555555, false);   // word 0
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0x55555555, true);    // word 1
    out[2 * i] = lo;
    out[2 * i + 1] = hi;
  }
}

static void dump(const char* name, const int* v, int n) {
  printf("\"%s\": [", name);
  for (int i = 0; i < n; ++i) printf("%s%u", i ? "," : "", (unsigned)v[i]);
  printf("]");
}

int main() {
  srand(7);
  int ha[512], hb[512], hsa[64], hsb[64];
  auto rnd_byte = []() {
    unsigned x = rand() & 0xFF;
    if ((x & 0x7F) == 0x7F) x ^= 1;  // no NaN codes
    if ((x & 0x78) == 0x78) x ^= 0x40;  // keep magnitudes moderate (exponent < 15)
    return x;
  };
  for (int i = 0; i < 512; ++i) {
    unsigned wa = 0, wb = 0;
    for (int b = 0; b < 4; ++b) {
      wa |= rnd_byte() << (8 * b);
      wb |= rnd_byte() << (8 * b);
    }
    ha[i] = (int)wa;
    hb[i] = (int)wb;
  }
  int *da, *db, *dsa, *dsb;
  float* dd;
  hipMalloc(&da, sizeof(ha));
  hipMalloc(&db, sizeof(hb));
  hipMalloc(&dsa, sizeof(hsa));
  hipMalloc(&dsb, sizeof(hsb));
  hipMalloc(&dd, 1024 * sizeof(float));
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
  hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  float hd[1024];
  for (int mode = 0; mode < 4; ++mode) {
    for (int l = 0; l < 64; ++l) {
      // E8M0 bytes: 127 = 1.0.  Distinct per lane (and per byte) so that the host can tell which lane's / byte's scale reached which output
      const unsigned e0 = 120 + (l % 13), e1 = 118 + (l % 11), e2 = 125 + (l % 5), e3 = 122 + (l % 7);
      const unsigned per_lane = e0 | (e1 << 8) | (e2 << 16) | (e3 << 24);
      hsa[l] = (mode == 1 || mode == 3) ? (int)per_lane : 0x7F7F7F7F;
      hsb[l] = (mode == 2) ? (int)per_lane : 0x7F7F7F7F;
    }
    hipMemcpy(dsa, hsa, sizeof(hsa), hipMemcpyHostToDevice);
    hipMemcpy(dsb, hsb, sizeof(hsb), hipMemcpyHostToDevice);
    if (mode == 3)
      hipLaunchKernelGGL((probe<1, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
    else
      hipLaunchKernelGGL((probe<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
    hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
    printf("{\"probe\": \"mfma_scale_f32_32x32x64 fp8 x fp8\", \"mode\": %d, ", mode);
    dump("a_regs", ha, 512);
    printf(", ");
    dump("b_regs", hb, 512);
    printf(", ");
    dump("scale_a", hsa, 64);
    printf(", ");
    dump("scale_b", hsb, 64);
    printf(", \"d\": [");
    for (int i = 0; i < 1024; ++i) printf("%s%.9g", i ? "," : "", hd[i]);
    printf("]}\n");
  }
  // ---- v_cvt_pk_fp8_f32
  float xs[64] = {0.f, 1.f, -1.f, 0.5f, 1.0625f, 1.125f, 1.1875f, 1.3125f, 448.f, 449.f, 464.f, 480.f, 500.f, 1000.f, 1e6f, -1e6f,
                  0.015625f, 0.0078125f, 0.001953125f, 0.0009765625f, 0.00048828125f, 3e-4f, 1e-4f, -3e-4f, 17.f, 18.f, 19.f, 21.f, 240.f, 256.f, 416.f, 432.f,
                  0.3f, 0.7f, 1.9f, 2.5f, 3.5f, 4.5f, 5.5f, 6.5f, 100.f, 200.f, 300.f, 400.f, -448.f, -449.f, 1e-8f, -1e-8f};
  for (int i = 48; i < 64; ++i) xs[i] = (float)(rand() % 2000 - 1000) / 37.f;
  float* dx;
  int* dout;
  int hout[64];
  hipMalloc(&dx, sizeof(xs));
  hipMalloc(&dout, sizeof(hout));
  hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, dx, 64, dout);
  hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
  printf("{\"probe\": \"v_cvt_pk_fp8_f32\", \"x\": [");
  for (int i = 0; i < 64; ++i) printf("%s%.9g", i ? "," : "", xs[i]);
  printf("], ");
  dump("out", hout, 64);
  printf("}\n");
  return 0;
}
