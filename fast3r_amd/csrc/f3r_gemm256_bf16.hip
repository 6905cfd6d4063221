// bf16 instantiations of the 256-tile GEMM kernel (f3r_gemm256_impl.h) + the measurement-only lab variants (tools/lab builds).
#include "f3r_gemm256_impl.h"

int f3r_gemm256_run_bf16(const f3r_gemm_args& a, hipStream_t stream, int stagger) { return dispatch256<BF16>(a, stream, stagger); }

// Measurement builds only: kernel_sel = 16 + LAB bits (bf16, plain operand, generic epilogue, no additive terms)
int f3r_gemm256_lab(const f3r_gemm_args& a, hipStream_t stream) {
#ifdef F3R_GEMM_LAB
  switch (a.kernel_sel - 16) {
#define F3R_LAB_CASE(n) case n: return launch_lab<BF16, n>(a, stream);
    F3R_LAB_CASE(0) F3R_LAB_CASE(1) F3R_LAB_CASE(2) F3R_LAB_CASE(3) F3R_LAB_CASE(4) F3R_LAB_CASE(7) F3R_LAB_CASE(8) F3R_LAB_CASE(16)
    F3R_LAB_CASE(32) F3R_LAB_CASE(33) F3R_LAB_CASE(64) F3R_LAB_CASE(96) F3R_LAB_CASE(128) F3R_LAB_CASE(256) F3R_LAB_CASE(512) F3R_LAB_CASE(384) F3R_LAB_CASE(640)
#undef F3R_LAB_CASE
  }
#endif
  f3r_set_error("f3r_gemm: kernel_sel %d is not available in this build", a.kernel_sel);
  return F3R_ERR_ARG;
}

