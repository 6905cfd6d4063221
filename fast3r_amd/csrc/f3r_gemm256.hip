// f3r_gemm256: the large-shape path of f3r_gemm -- host-side selection + the fp16 instantiations of f3r_gemm256_impl.h
// (the bf16 ones live in f3r_gemm256_bf16.hip so that the two halves compile in parallel).
#include "f3r_gemm256_impl.h"

int f3r_gemm256_run_bf16(const f3r_gemm_args& a, hipStream_t stream, int stagger);  // f3r_gemm256_bf16.hip
int f3r_gemm256_run_conv_f8_fin(const f3r_gemm_args& a, hipStream_t stream);         // f3r_gemm256_f8.hip

// Whether the 256-tile kernel takes this (already validated) problem: everything its LDS-DMA staging cannot express -- K tails, ragged
// channel counts, strided or pre-activated conv operands, small or narrow outputs -- stays on the 128-tile kernel.
bool f3r_gemm256_eligible(const f3r_gemm_args& a) {
  if (a.N % 128 != 0) return false;
  if (a.epi == F3R_EPI_QKV) {  // a 256-wide tile must lie in ONE of the q / k / v parts
    const int Dq = a.qkv_dq ? a.qkv_dq : a.N / 3;
    if (Dq % 256 != 0 || ((a.N - Dq) / 2) % 256 != 0) return false;
  }
  const int Kpad1 = a.split ? a.Kpad / 2 : a.Kpad;
  if (Kpad1 % 64 != 0) return false;
  if (a.a_mode == F3R_A_PLAIN) {
    if (a.K != Kpad1) return false;
    if ((int64_t)256 * a.lda * 2 >= (1ll << 32)) return false;
  } else {
    if (a.a_relu || a.conv_C % 64 != 0) return false;
    // 32-bit byte offsets into the NHWC operand (B * H * W pixels; stride 2 reads four times the output's pixel count)
    if (a.M / ((int64_t)a.conv_OH * a.conv_OW) * a.conv_H * a.conv_W * a.conv_C * 2 >= (1ll << 32)) return false;
    if (a.split == F3R_SPLIT_X3F8 && (a.conv_C % 128 != 0 || a.dtype != F3R_F16)) return false;
  }
  if (a.split == F3R_SPLIT_X3F8 && a.a_mode != F3R_A_CONV3X3) return false;
  if (a.fin_w && (a.a_mode != F3R_A_CONV3X3 || a.N != 128 || a.epi != F3R_EPI_GENERIC)) return false;
  if ((int64_t)256 * a.Kpad * 2 >= (1ll << 32)) return false;
  // additive epilogue terms enter through the accumulators: one kind at a time, no activation in between, and only the kinds the
  // model uses on each operand mode (fp32 residual / image-id rows on plain GEMMs, lowp skip connections on convolutions)
  const int add = gemm_additive_pattern(a);
  if (add == F3R_ADD_UNSUPPORTED || (add != F3R_ADD_NONE && a.act != F3R_ACT_NONE)) return false;
  if (add != F3R_ADD_NONE && a.epi != F3R_EPI_GENERIC) return false;
  if (a.a_mode == F3R_A_CONV3X3 ? (add == F3R_ADD_RES_F32 || add == F3R_ADD_ROWADD) : add == F3R_ADD_RES_LP) return false;
  return true;
}

// Whether the 256-tile kernel (in the tile form tile_halves picks) is also the FASTER choice: one workgroup per CU, so with few tiles or
// an unlucky tail round the 128-tile kernel (2 workgroups per CU, 4x the tiles) fills the chip better -- see tile_score.
bool f3r_gemm256_preferred(const f3r_gemm_args& a) { return score_256(a, tile_halves(a)) >= score_128(a); }

int f3r_gemm256_launch(const f3r_gemm_args& a, hipStream_t stream, int stagger) {
  const int64_t tiles = ((a.M + BM - 1) / BM) * ((a.N + 127) / 128);
  if (tiles <= 0) return F3R_OK;
  F3R_REQUIRE(tiles < (1ll << 31), "f3r_gemm: grid too large");
  if (a.split == F3R_SPLIT_X3F8 || a.fin_w) return f3r_gemm256_run_conv_f8_fin(a, stream);
  return a.dtype == F3R_F16 ? dispatch256<F16>(a, stream, stagger) : f3r_gemm256_run_bf16(a, stream, stagger);
}
