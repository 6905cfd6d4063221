// f3r_gemm256_f8: the round-6 instantiations of the 256-tile kernel (f3r_gemm256_impl.h) for the DPT head's 3x3 convolutions --
//   F8  (F3R_SPLIT_X3F8): the two correction products of X3 on the block-scaled fp8 MFMA (fp16 high planes);
//   FIN (f3r_gemm_args.fin_w): head[2]'s epilogue carries ReLU + head[4] (1x1 conv to 3 / 4 channels) + postprocess (gemm_epilogue_fin).
// A translation unit of its own so that it compiles beside f3r_gemm256.hip / f3r_gemm256_bf16.hip.
#include <cstdlib>

#include "f3r_gemm256_impl.h"

// (already validated by f3r_gemm and found eligible by f3r_gemm256_eligible)
// kernel_sel 5 on the fused-tail x3f8 launch (measurement only): the MERGED schedule of f3r_gemm256_impl.h -- 32 MFMAs per barrier pair.  Measured and
// NOT taken: head[2] + tail 4.52 ms against 4.15 ms (profiles/r06_conv_merged_phases_*_rejected.jsonl; `tools/conv_f8_ab.py --splits x3f8,x3f8:5`).
int f3r_gemm256_run_conv_f8_fin(const f3r_gemm_args& a, hipStream_t stream) {
  const bool f8 = a.split == F3R_SPLIT_X3F8;
  const int add = gemm_additive_pattern(a);
  if (a.fin_w) {
#ifdef F3R_CONV_ABLATIONS
    // tools/lab builds only (-DF3R_CONV_ABLATIONS, F3R_CONV_ABLATE=<bits> in the environment): the fused-tail x3f8 launch with parts of its loop removed
    // -- TIMING ONLY, the results are garbage: 1 no LDS-DMA, 1024 no W stream, 2048 no A stream, 2 no fragment reads, 4 no MFMAs, 7 barriers only
    if (f8) {
      static const char* ab = getenv("F3R_CONV_ABLATE");
      switch (ab ? atoi(ab) : 0) {
#define F3R_ABL(n) case n: return launch256<F16, F3R_A_CONV3X3, F3R_EPI_GENERIC, 1, F3R_ADD_NONE, 1, true, true, false, n>(a, stream);
        F3R_ABL(1) F3R_ABL(1024) F3R_ABL(2048) F3R_ABL(2) F3R_ABL(4) F3R_ABL(3) F3R_ABL(7)
#undef F3R_ABL
      }
    }
#endif
    if (a.dtype == F3R_BF16) return launch256<BF16, F3R_A_CONV3X3, F3R_EPI_GENERIC, 1, F3R_ADD_NONE, 1, false, true>(a, stream);
    if (f8 && a.kernel_sel == 5) return launch256<F16, F3R_A_CONV3X3, F3R_EPI_GENERIC, 1, F3R_ADD_NONE, 1, true, true, true>(a, stream);
    return f8 ? launch256<F16, F3R_A_CONV3X3, F3R_EPI_GENERIC, 1, F3R_ADD_NONE, 1, true, true>(a, stream)
              : launch256<F16, F3R_A_CONV3X3, F3R_EPI_GENERIC, 1, F3R_ADD_NONE, 1, false, true>(a, stream);
  }
  const int nh = tile_halves(a);
  if (add == F3R_ADD_RES_LP)
    return nh == 2 ? launch256<F16, F3R_A_CONV3X3, F3R_EPI_GENERIC, 1, F3R_ADD_RES_LP, 2, true>(a, stream)
                   : launch256<F16, F3R_A_CONV3X3, F3R_EPI_GENERIC, 1, F3R_ADD_RES_LP, 1, true>(a, stream);
  return nh == 2 ? launch256<F16, F3R_A_CONV3X3, F3R_EPI_GENERIC, 1, F3R_ADD_NONE, 2, true>(a, stream)
                 : launch256<F16, F3R_A_CONV3X3, F3R_EPI_GENERIC, 1, F3R_ADD_NONE, 1, true>(a, stream);
}
