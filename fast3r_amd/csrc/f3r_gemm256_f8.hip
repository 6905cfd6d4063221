// f3r_gemm256_f8: the round-6 instantiations of the 256-tile kernel (f3r_gemm256_impl.h) for the DPT head's 3x3 convolutions --
//   F8  (F3R_SPLIT_X3F8): the two correction products of X3 on the block-scaled fp8 MFMA (fp16 high planes);
//   FIN (f3r_gemm_args.fin_w): head[2]'s epilogue carries ReLU + head[4] (1x1 conv to 3 / 4 channels) + postprocess (gemm_epilogue_fin).
// A translation unit of its own so that it compiles beside f3r_gemm256.hip / f3r_gemm256_bf16.hip.
#include "f3r_gemm256_impl.h"

// (already validated by f3r_gemm and found eligible by f3r_gemm256_eligible)
// kernel_sel 5 on the fused-tail x3f8 launch (measurement only): the MERGED schedule of f3r_gemm256_impl.h -- 32 MFMAs per barrier pair.  Measured and
// NOT taken: head[2] + tail 4.52 ms against 4.15 ms (profiles/r06_conv_merged_phases_*_rejected.jsonl; `tools/conv_f8_ab.py --splits x3f8,x3f8:5`).
int f3r_gemm256_run_conv_f8_fin(const f3r_gemm_args& a, hipStream_t stream) {
  const bool f8 = a.split == F3R_SPLIT_X3F8;
  const int add = gemm_additive_pattern(a);
  if (a.fin_w) {
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
