/*
 * f3r.h -- C ABI of libf3r_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the Fast3R
 * single-forward-pass inference hot path
 *     fast3r/dust3r/inference_multiview.py:70-99  inference()
 *       -> fast3r/models/fast3r.py:302-497        Fast3R.forward
 *
 * The reference has exactly one native binding, the pybind `curope.rope_2d(tokens, positions,
 * base, fwd)` (fast3r/croco/models/curope/curope.cpp:49-69); everything else it computes through
 * torch ops (nn.Linear / Conv2d / LayerNorm / SDPA / F.interpolate).  This header is what a
 * from-scratch native backend for the same path binds instead: one entry point per fused op,
 * plain device pointers + sizes, no torch types.  Each entry cites the reference code it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (the PyTorch-ROCm allocator); the
 *     library allocates nothing and keeps no mutable state except the calling thread's last-error
 *     string (kernel choices that used to be process-wide knobs are per-call fields: f3r_gemm_args.kernel_sel);
 *   - every launch is asynchronous on the caller's `stream` (unlike the reference extension,
 *     which launches on the legacy default stream: curope/kernels.cu:102);
 *   - return value: F3R_OK (0) or a negative f3r_status; nothing throws across the boundary.
 *     The Python host (fast3r_amd/_lib.py) maps errors onto the exception types the reference
 *     raises (ValueError / AssertionError / RuntimeError);
 *   - "lowp" = the 16-bit MFMA operand type selected by `dtype` (F3R_F16 or F3R_BF16);
 *     accumulation, residual stream, LayerNorm statistics, softmax and post-processing are fp32.
 */
#ifndef F3R_H_
#define F3R_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* f3r_stream_t; /* hipStream_t */

typedef enum {
  F3R_OK = 0,
  F3R_ERR_ARG = -1,         /* bad argument (null pointer, bad size, misalignment) */
  F3R_ERR_UNSUPPORTED = -2, /* shape / mode not implemented by the kernels */
  F3R_ERR_LAUNCH = -3       /* hipGetLastError() after the launch was not hipSuccess */
} f3r_status;

typedef enum { F3R_F16 = 0, F3R_BF16 = 1 } f3r_dtype;

#define F3R_MAX_SEG 8

/* library version (major*10000 + minor*100 + patch) and last error text of the calling thread */
int f3r_version(void);  /* 350 = 0.3.5, round 6 (F3R_SPLIT_X3F8, f3r_gemm_args.out_f8 / out_relu_f8 / fin_*, f3r_interp_bilinear_f8); 340 = 0.3.4, round 6 (+ f3r_block_workspace_bytes_ex; the library clears sched_counter per launch); 330 = 0.3.3, round 5 (f3r_attn_args.dbg_counters is uint32[8] incl. two clock sums; f3r_wall_clock_khz); 320 = 0.3.2, round 4 (+ f3r_attn_f32_mfma, head_dim 80 / 128 kernels); 310: f3r_attn_args.dbg_counters, f3r_gemm_args.kernel_sel 6; 300 = round 3; 200 = round 2 */
const char* f3r_last_error_string(void);
/* sizeof(f3r_gemm_args) (what == 0) / sizeof(f3r_attn_args) (what == 1) / sizeof(f3r_attn_f32_args) (what == 2): lets a foreign-language binding
   verify its struct layout before the first call; 0 for an unknown `what` */
size_t f3r_sizeof(int what);
/* rate of the constant clock s_memrealtime counts on the current device, in kHz (hipDeviceAttributeWallClockRate; 100 000 on MI355X); 0 on failure */
int f3r_wall_clock_khz(void);

/* ---------------------------------------------------------------------------------------------
 * f3r_patchify: fp32 NCHW image -> lowp im2col rows for the k=s=ps patch-embedding convolution.
 * Replaces the input side of PatchEmbedDust3R.forward's Conv2d (fast3r/dust3r/patch_embed.py:24-38,
 * fast3r/croco/models/blocks.py:412-415) and of DINOv2's PatchEmbed (DinoEncoder, fast3r/models/fast3r.py:561-651, ps = 14).
 * out[(b*h+py)*w+px][c*ps*ps+dy*ps+dx] = img[b][c][py*ps+dy][px*ps+dx] (the Conv2d weight's own (c,dy,dx) order, so
 * weight.view(D, 3*ps*ps) is the GEMM operand).  ld_out = row stride of `out` in elements (0 = 3*ps*ps; a multiple of 8;
 * columns >= 3*ps*ps are written as zeros: 3*14*14 = 588 is padded to 640 this way).
 */
int f3r_patchify(const float* img, void* out, int batch, int H, int W, int ps, int ld_out, int dtype, f3r_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * f3r_layernorm: rows of fp32 -> normalised lowp (GEMM operand) and/or fp32.
 * Replaces nn.LayerNorm in Block.forward (blocks.py:236-239), enc_norm (fast3r.py:558) and dec_norm
 * (fast3r.py:805).  Biased variance, y=(x-mu)/sqrt(var+eps)*gamma+beta; eps is 1e-6 or 1e-5 per
 * site (fast3r.py:509,683,700).  rms!=0: RMSNorm (no mean, no beta: components/llama.py:137-163).
 * D % 4 == 0.  out_lp / out_f32 may each be NULL.
 */
int f3r_layernorm(const float* x, const float* gamma, const float* beta, void* out_lp, float* out_f32,
                  int64_t rows, int D, float eps, int rms, int dtype, f3r_stream_t stream);
/* the same LayerNorm / RMSNorm writing the operand rows of F3R_SPLIT_W2F8 (f3r_split): out row r = [D fp16 | D fp8 e4m3 of the same values
   clamped to +-448], ld_out sixteen-bit elements apart (>= 3 D / 2, a multiple of 8).  fp16 only; D % 8 == 0. */
int f3r_layernorm_f8(const float* x, const float* gamma, const float* beta, void* out_rows, int64_t ld_out, int64_t rows, int D, float eps, int rms,
                     f3r_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * f3r_gemm: out = epilogue( A(M,K) * W(N,K)^T ) on MFMA, fp32 accumulate.
 * Replaces every nn.Linear on the path (blocks.py:94-105 Mlp, :125-131,138,169 Attention qkv/proj,
 * fast3r.py:782 decoder_embed), the patch-embed Conv2d (as GEMM over f3r_patchify rows), and all
 * Conv2d / ConvTranspose2d of the DPT head (dpt_block.py:42-78,105-154,187-250,365-382,416-481).
 */
typedef enum { F3R_A_PLAIN = 0, F3R_A_CONV3X3 = 1 } f3r_a_mode;
typedef enum { F3R_EPI_GENERIC = 0, F3R_EPI_QKV = 1, F3R_EPI_CONVT = 2 } f3r_epi_mode;
typedef enum { F3R_ACT_NONE = 0, F3R_ACT_GELU = 1, F3R_ACT_RELU = 2 } f3r_act;

typedef struct f3r_gemm_args {
  /* operands */
  const void* A;     /* lowp. PLAIN: [M][lda].  CONV3X3: NHWC [B][conv_H][conv_W][conv_C] */
  const void* W;     /* lowp [N][Kpad] row-major, zero padded. PLAIN: k < K real.  CONV3X3: k = tap*Cpad + ci */
  const float* bias; /* [N] or NULL */
  int64_t M;         /* rows (tokens / output pixels) */
  int32_t N;
  int32_t K;         /* PLAIN: real K (multiple of 8).  CONV3X3: ignored */
  int32_t Kpad;      /* multiple of 64.  CONV3X3: 9*Cpad with Cpad = roundup(conv_C, 64) */
  int64_t lda;       /* PLAIN: row stride of A in elements */
  int32_t a_mode;    /* f3r_a_mode */
  int32_t a_relu;    /* ReLU applied to A as it is staged (the pre-activation of ResidualConvUnit_custom, dpt_block.py:143,148) */
  int32_t conv_H, conv_W, conv_C, conv_stride, conv_OH, conv_OW; /* CONV3X3 (pad 1): M = B*conv_OH*conv_OW */
  /* epilogue */
  int32_t epi;       /* f3r_epi_mode */
  int32_t act;       /* f3r_act, applied after bias */
  const float* rowadd; /* GENERIC: out += rowadd[m / rowadd_div][n] (image-index embedding, fast3r.py:799) or NULL */
  int64_t rowadd_div;
  const float* res_f32; /* GENERIC: fp32 residual [M][ldr_f32] or NULL (x + attn(..), x + mlp(..): blocks.py:237-238) */
  int64_t ldr_f32;
  const void* res_lp;   /* GENERIC: lowp residuals (skip_add of the RCU / fusion block, dpt_block.py:154,216) or NULL */
  int64_t ldr_lp;
  const void* res_lp2;
  int64_t ldr_lp2;
  float* out_f32;    /* GENERIC: fp32 output [M][ldo_f32] or NULL */
  int64_t ldo_f32;
  void* out_lp;      /* GENERIC: lowp output [M][ldo_lp] or NULL.  CONVT: NHWC [B][h*s][w*s][ct_cout] */
  int64_t ldo_lp;
  /* QKV epilogue (N = 3*D, D = heads*64): q -> [M][D], k -> [M][D], v -> transposed vt[m / seq_len][D][ldvt] at column m % seq_len */
  void* q;
  void* k;
  void* vt;
  int64_t seq_len;
  int64_t ldvt;
  const float* rope_cos; /* [n_pos][16] fp32 cos/sin of pos * 100^(-i/16) (pos_embed.py:139-150) or NULL = no RoPE (fusion decoder) */
  const float* rope_sin;
  int32_t rope_w;    /* tokens per image row: token p of a view sits at (y, x) = (p / rope_w, p % rope_w) (blocks.py:376-388) */
  float q_scale;     /* QKV: q (after bias and RoPE) is multiplied by this before rounding; 0 means 1.  The attention kernel
                        wants softmax_scale*log2(e) folded in here (f3r_attn_args.q_prescaled) */
  /* CONVT epilogue (ConvTranspose2d with kernel == stride == ct_s): rows m = (b, y, x) on a ct_h x ct_w grid,
     columns n = (dy*ct_s + dx)*ct_cout + co  ->  out_lp[b][y*ct_s+dy][x*ct_s+dx][co] */
  int32_t ct_s, ct_h, ct_w, ct_cout;
  int32_t dtype;     /* f3r_dtype */
  int32_t rope_mode; /* QKV with rope_cos != NULL.  0: RoPE-2D of the CroCo encoder as described above.  1: one rotation angle set per
                        ROW GROUP (LlamaDecoder, fast3r/models/components/llama.py:96-122 with freqs_cis gathered per view,
                        fast3r.py:872-922): rope_cos / rope_sin are [n_groups][32] (32 complex pairs of a 64-wide head), row m uses
                        group m / rope_w (rope_w = tokens per view, or 1 with one table row per token); the first 32 dims of a head
                        rotate with table columns 0-15 (dim i pairs with i+16), the last 32 with columns 16-31 -- the host permutes
                        the q / k weight rows so that the reference's interleaved pairs (2j, 2j+1) land on these positions */
  /* Split-precision operands (the "high" precision mode, DESIGN.md section 3 (Precision modes)): a value x is carried as two lowp numbers
     hi = lowp(x), lo = lowp(x - hi) (~22 significand bits with fp16 pieces) and the product is summed over K SEGMENTS on the same
     MFMA path (fp32 accumulate):  F3R_SPLIT_W2: A.W_hi + A.W_lo (weights exact to ~2^-22, activations single);
     F3R_SPLIT_X3: A_hi.W_hi + A_hi.W_lo + A_lo.W_hi.  With split != 0, W is [N][2][Kpad/2] (plane 0 = hi, plane 1 = lo; Kpad is still
     the row stride, K the real depth of ONE plane) and, for X3, A_lo is the low plane of A (same layout / strides as A). */
  int32_t split;     /* f3r_split */
  int32_t kernel_sel; /* 0 = pick the kernel by shape; 1 = 128x128-tile kernel; 2 / 3 = 256x256-tile kernel with / without staggered wave rows;
                         4 = its 256x128 tile form; 5 = 256x256 with one tile per workgroup instead of the persistent grid (2 - 5 are for
                         measurement: an ineligible shape is F3R_ERR_ARG, never a silent fallback);
                         6 = the hand-scheduled one-wave-per-SIMD kernel (csrc/asm/gemm_gen.py: 4 waves x 128 x 128 outputs of
                         v_mfma_f32_32x32x16, five-slot LDS-DMA ring; ABI 310).  It takes plain GEMMs with the GENERIC epilogue, M and N multiples
                         of 256, K == Kpad (split NONE or W2), ONE output: fp32 (+ bias, + fp32 residual, no activation) or lowp (+ bias,
                         + GELU / ReLU), or F3R_EPI_QKV without rotary embedding and equal q / k / v widths (two launches) -- F3R_ERR_UNSUPPORTED otherwise.
                         0 uses it for eligible launches whose 256 x 256 tiles fill its persistent grid (>= one tile per CU, last round >= 80 % full);
                         7 = pick by shape among the compiler-scheduled kernels only;
                         9 = like 6 but WITH a start-up skew of the persistent workgroups (measurement only, round 5: workgroup b of a launch with >= 2
                         tiles per workgroup starts ((b / 8) % 4) quarter tile periods late, so that the epilogues' HBM traffic of one quarter of the
                         chip overlaps the K loops of the rest; measured 6 % SLOWER on fc2, neutral elsewhere -- lock-step workgroups share their
                         operand panels through L2 -- hence not the default) */
  const void* A_lo;
  /* low planes of the lowp outputs / residuals (NULL = not carried): out_lp_lo = lowp(v - float(out_lp)); res_lp*_lo are added like
     their high planes.  Same leading dimensions as the high planes. */
  void* out_lp_lo;
  const void* res_lp_lo;
  const void* res_lp2_lo;
  /* GENERIC: second lowp output relu(v) [M][ldo_lp] (+ its low plane): the pre-activated copy the next ResidualConvUnit conv reads
     (dpt_block.py:143,148), so that conv needs no a_relu and can stage its operand by LDS-DMA */
  void* out_relu;
  void* out_relu_lo;
  /* QKV epilogue with grouped-query attention (LlamaDecoder n_kv_heads < n_heads, components/llama.py:195-198,220-232): the N columns are
     [ q: qkv_dq | k: (N - qkv_dq) / 2 | v: (N - qkv_dq) / 2 ], q -> [M][qkv_dq], k -> [M][Dkv], v -> vt[m / seq_len][Dkv][ldvt].
     0 = three equal thirds (N / 3 each).  Both widths must be multiples of 64 (whole heads). */
  int32_t qkv_dq;
  /* F3R_SPLIT_W2F8 with act = GELU and out_lp (ABI 330; was reserved): != 0 = the rows of out_lp are [N fp16 | N fp8] (ldo_lp >= 3 N / 2) and the
     epilogue also writes the fp8 (e4m3, clamped to 448) copy of its output N sixteen-bit elements into the row: the A operand of the NEXT
     W2F8 GEMM (fc1 -> fc2 of a transformer MLP, blocks.py:94-105) */
  int32_t out_lp_f8;
  const uint32_t* w_scale; /* F3R_SPLIT_W2F8: [N] scale words of the weight rows' fp8 plane (see f3r_split); NULL otherwise (ABI 330) */
  /* F3R_EPI_QKV with F3R_SPLIT_W2F8 (ABI 330): W / w_scale hold the q and k rows only ([N - Dkv] rows of [K fp16 | K fp8]); the v rows come as TWO
     fp16 planes here ([Dkv][2 K], as F3R_SPLIT_W2 packs them): the V^T launch runs with swapped operand roles on the fp16 kernel (its "weights"
     are the activations).  NULL otherwise. */
  const void* W_aux;
  /* ABI 350 (round 6).  fp8 COPIES of the lowp outputs for the next F3R_SPLIT_X3F8 convolution (GENERIC epilogue; NULL = not written): row m of
     out_f8 is 2 N bytes [ e4m3(clamp(v, +-448)) x N | e4m3(clamp((v - float(fp16(v))) 2^12, +-448)) x N ] for the values v that out_lp rounds;
     out_relu_f8 the same for relu(v) (what out_relu carries).  fp16 only, N % 8 == 0, 8-byte aligned. */
  void* out_f8;
  void* out_relu_f8;
  /* ABI 350: the tail of the DPT head fused into the epilogue of its last 3x3 convolution (head[2] + ReLU -> head[4] 1x1 conv -> postprocess,
     dpt_block.py:375-381, heads/postprocess.py:16-64): with fin_w != NULL the launch must be a CONV3X3 whose N <= 128 output channels lie in ONE
     256 x 128 tile of the 256-tile kernel (N % 128 == 0 today: the DPT head's last_dim = 128); the epilogue applies `act` to the fp32
     accumulators, multiplies them by fin_w (fp32 [4][N], rows >= fin_n_out zero) + fin_b (fp32 [4]) on the vector pipe and writes
     fin_pts (fp32 [M][3]) / fin_conf (fp32 [M] or NULL) exactly as f3r_dpt_final does (fin_depth_mode / fin_conf_mode / fin_vmin / fin_vmax: its
     arguments).  No lowp output is written then (out_lp may be NULL): the 128-channel activation never reaches HBM.  F3R_ERR_UNSUPPORTED when the
     launch cannot take the 256 x 128 tile form. */
  const float* fin_w;
  const float* fin_b;
  float* fin_pts;
  float* fin_conf;
  int32_t fin_n_out, fin_depth_mode, fin_conf_mode;
  float fin_vmin, fin_vmax;
  int32_t reserved0;
} f3r_gemm_args;

/* F3R_SPLIT_W2F8 (ABI 330): W2 with the LOW plane of the weights, and the copy of the activations it multiplies, in fp8 (OCP e4m3) on the
   block-scaled MFMA of gfx950 (v_mfma_scale_f32_32x32x64_f8f6f4: 1.40x the matrix-pipe rate of the two-fp16-plane form under the power cap,
   profiles/r05_ubench_mfma_mixed_fp16_fp8_fp6.jsonl): A(M, K) rows are [K fp16 | K fp8] (lda >= 3 K / 2 sixteen-bit elements; the fp8 copy holds
   the same numbers clamped to +-448: f3r_layernorm_f8 writes such rows), W rows are [K fp16 hi | K fp8 e4m3((W - hi) 2^s_n)] (row stride 3 K
   bytes: Kpad = K, a multiple of 128) with one power-of-two scale per output channel in w_scale (E8M0 byte 127 - s_n replicated into the four
   bytes of a word).  The correction term A W_lo is 2^-11 of the product and tolerates the 2^-4 relative error of both fp8 operands: the
   weight's rounding error drops ~24x instead of vanishing (tools/emu_gemm.py run_case_f8).  fp16 operands, plain A, GENERIC epilogue with ONE
   output (as kernel_sel 6), M and N multiples of 256 -- anything else is F3R_ERR_UNSUPPORTED (there is no second kernel for this layout). */
/* F3R_SPLIT_X3F8 (ABI 350; the 3x3 convolutions of the DPT head, dpt_block.py:133-154,365-382): X3 whose two CORRECTION products run on the
   block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4): A_hi W_hi on fp16 + A_hi8 W_lo8 + A_lo8 W_hi8 -- two matrix-pipe units instead of
   three (each correction is 2^-11 of the product and tolerates the 2^-4 relative error of both fp8 operands; oracle/precision_study.py --study
   heads_f8: the stress fixture moves from 6.94e-4 to 6.93e-4).  CONV3X3 only, stride 1, conv_C % 128 == 0, fp16, the 256-tile kernel
   (F3R_ERR_UNSUPPORTED otherwise).  A = the fp16 high plane (NHWC); A_lo = the fp8 planes of the same tensor, per pixel [C bytes e4m3(hi, clamped to
   +-448) | C bytes e4m3(lo 2^12, clamped)] (what out_f8 / f3r_interp_bilinear_f8 write); W rows are [9 C fp16 hi | 9 C bytes e4m3(w_lo 2^s_n) |
   9 C bytes e4m3(w_hi 2^t_n)] (4 * 9 C bytes: Kpad = 2 * 9 C as for X3, k = tap * C + ci in every plane) and w_scale[n] holds the E8M0 bytes
   127 - s_n (byte 0) and 127 - t_n (byte 1) (ops.pack_conv3x3_weight_f8). */
typedef enum { F3R_SPLIT_NONE = 0, F3R_SPLIT_W2 = 1, F3R_SPLIT_X3 = 2, F3R_SPLIT_W2F8 = 3, F3R_SPLIT_X3F8 = 4 } f3r_split;

int f3r_gemm(const f3r_gemm_args* args, f3r_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * f3r_block_workspace_bytes: size and layout of the intermediates of ONE transformer block (Block.forward, blocks.py:236-239) over
 * `tokens` rows = n_seq sequences of seq_len tokens, model width D, MLP hidden width `hidden` (2 x the SwiGLU width for the
 * LlamaDecoder's stacked [w1; w3] projection).  The library allocates nothing: the caller makes ONE allocation of the returned size
 * per encoder pass / decoder sample and every block of that pass reuses it (the intermediates of a block are dead when it ends).
 * offsets[0..4] (bytes, 256-byte aligned) = { LN output / attention output (lowp [tokens][D]),  q (lowp [tokens][D]),
 * k (lowp [tokens][kv_dim]),  V^T (lowp [n_seq][kv_dim][ldvt], ldvt = seq_len rounded up to 64: pad columns are never written, zero
 * them once; kv_dim = D, or n_kv_heads * 64 with grouped-query attention, or 0 when K / V^T live elsewhere -- the view-sharded path
 * writes them into the exchange buffers -- and the two regions are empty),
 * MLP hidden (lowp [tokens][hidden]) }.  Returns 0 on bad arguments.
 */
size_t f3r_block_workspace_bytes(int64_t tokens, int D, int kv_dim, int hidden, int64_t n_seq, int64_t seq_len, size_t offsets[5]);
/* ABI 340: the same layout for passes that run the F3R_SPLIT_W2F8 GEMMs (f8_rows != 0): regions 0 and 4 are sized for rows [w fp16 | w fp8]
   (3 w bytes per row, w = D / hidden; both multiples of 8) -- what f3r_layernorm_f8 and the GELU epilogue (out_lp_f8) write -- and the plain
   [tokens][w] 16-bit forms of the same intermediates (attention output; a hidden state kept on fp16 planes) alias the head of those regions: the
   wide and the plain form of a region are never live at the same time inside a block.  f8_rows = 0 is f3r_block_workspace_bytes. */
size_t f3r_block_workspace_bytes_ex(int64_t tokens, int D, int kv_dim, int hidden, int64_t n_seq, int64_t seq_len, int f8_rows, size_t offsets[5]);

/* ---------------------------------------------------------------------------------------------
 * f3r_attn_fwd: O = softmax(scale * Q K^T) V, head_dim 64 (other widths: f3r_attn_args.head_dim), flash-style (never forms
 * the T x T matrix), fp32 online softmax.  Replaces the q@k^T -> softmax -> @v core of
 * Attention.forward (blocks.py:158-190; all three `attn_implementation`s compute this).
 * K/V arrive as up to F3R_MAX_SEG segments so that the view-sharded multi-GPU path can attend
 * over the local shard plus the all-gathered remote shards without concatenating them.
 *   q  : [batch][tq][ldq]   (head h at columns h*64 .. h*64+63)
 *   k_seg[s]  : [batch][seg_len[s]][ldk]
 *   vt_seg[s] : [batch][heads*64][ldvt[s]]  (V transposed, as written by the F3R_EPI_QKV epilogue)
 *   o  : [batch][tq][ldo]
 */
typedef struct f3r_attn_args {
  const void* q;
  void* o;
  int64_t ldq, ldo;
  int64_t q_batch_stride, o_batch_stride; /* elements */
  int64_t tq;
  int32_t batch;
  int32_t n_heads;
  int32_t n_seg;
  int32_t dtype;
  const void* k_seg[F3R_MAX_SEG];
  const void* vt_seg[F3R_MAX_SEG];
  int64_t seg_len[F3R_MAX_SEG];
  int64_t ldvt[F3R_MAX_SEG];
  int64_t ldk;
  int64_t k_batch_stride[F3R_MAX_SEG];  /* elements */
  int64_t vt_batch_stride[F3R_MAX_SEG]; /* elements */
  float scale; /* 0.125, or 0.160192 for the fusion decoder in eval mode (blocks.py:119-124,151-154) */
  int32_t q_prescaled; /* != 0: q was already multiplied by scale*log2(e) (F3R_EPI_QKV with q_scale) before its one
                          rounding to lowp, so the kernel exponentiates with exp2 directly; `scale` is then unused */
  /* Online-softmax state carried between launches over different K/V segments of the SAME queries (view-sharded path:
     launch 1 attends over the local shard while the all-gather of the remote shards is in flight and writes the state,
     launch 2 resumes from it over the remote shards and writes the normalised output).
       st_o  : fp32 [batch][tq][n_heads*64]   un-normalised O accumulators
       st_ml : fp32 [batch][tq][n_heads][4]   {running max, partial row sum of lane half 0, of lane half 1, unused}
     state_in != 0: start from the state instead of (O=0, m=-inf, l=0).  state_out != 0: write the state and NOT o. */
  float* st_o;
  float* st_ml;
  int32_t state_in;
  int32_t state_out;
  /* Grouped-query attention (repeat_kv, components/llama.py:125-134,229-232): query head h reads K / V head h / kv_group; the K rows
     and V^T planes then hold n_heads / kv_group heads.  0 or 1 = one K / V head per query head. */
  int32_t kv_group;
  /* Causal attention (F.scaled_dot_product_attention(..., is_causal=True), components/llama.py:239): key at sequence position j is
     visible to the query at position i iff j <= i.  Positions are GLOBAL token indices: query row r of this launch sits at
     q_pos0 + r, key row r of segment s at seg_pos0[s] + r (the view-sharded path attends over shards that start anywhere). */
  int32_t causal;
  int64_t q_pos0;
  int64_t seg_pos0[F3R_MAX_SEG];
  /* Kernel choice (per call, like f3r_gemm_args.kernel_sel; no process-wide switch):
       0 = automatic: the hand-scheduled one-wave-per-SIMD kernel (csrc/asm/attn_gen.py: 512-query workgroups, 128 queries per
           wave; a partial last workgroup is fine) when the launch is eligible -- no causal mask, q_prescaled, tq >= 128, every non-empty K/V segment a
           multiple of 64 keys with one ldvt and one pair of batch strides, at least F3R_ATTN_ASM_MIN_KEYS keys in total, kv_group a
           power of two, batch 1 when the softmax state is carried (state_in / state_out; the state layout is the HIP kernel's, so
           the two kernels can resume each other's launches) -- and the general HIP kernel otherwise;
       1 = the general HIP kernel;  2 = the hand-scheduled kernel (F3R_ERR_UNSUPPORTED if the launch is not eligible). */
  int32_t kernel_sel;
  /* Width of a head: 0 or 64 = the tuned kernels; any other multiple of 16 up to 128 (the reference's Attention takes any dim //
     num_heads, blocks.py:113-143; its model_scaling_huge.yaml fusion decoder has 80): same layouts with head_dim columns per head
     (q / k / o rows, head_dim V^T planes per head, st_o rows of n_heads * head_dim), no causal mask.  80 and 128 have hand-scheduled
     kernels of their own (csrc/asm/attn_gen.py: 256-query workgroups, 64 queries per wave; eligibility as for kernel_sel 0 below
     with tq >= 64 and NO minimum number of keys: they beat the generic kernel from 128 keys on,
     profiles/r04_attn_head_dim_short_sequences.jsonl); everything else, and launches those kernels cannot take, run the generic kernel
     (f3r_attn_generic.hip). */
  int32_t head_dim;
  /* Optional counters of the hand-scheduled kernel (NULL = none; ABI 310, widened in ABI 330): device uint32[56], 8-byte aligned, zeroed by the
     caller; every wave of a launch that takes that kernel adds {entries into the block that moves the softmax reference (the forced first
     one included), 1} and, as three uint64 at bytes 8, 16 and 24, {64-key tiles it walked, shader-clock cycles (s_memtime) the wave lived,
     ticks of the constant-rate clock (s_memrealtime, f3r_wall_clock_khz) over the same span}, followed from byte 32 on by the
     same sums per XCD (8 records of three uint64: cycles, ticks, waves; the XCD is read from HW_REG_XCC_ID).  bench.py reports (entries - waves) / waves
     (how often the lazy reference really moved) and, from the clocks, the effective shader clock and the matrix-pipe utilisation of the
     timed launches themselves (roofline.live). */
  uint32_t* dbg_counters;
  /* Work stealing for the hand-scheduled kernels (NULL = off; ABI 330): device uint32[2] {next item, workgroups done}.  The library clears the two
     words on `stream` ahead of every launch that uses them (since round 6; the kernel also leaves them zero), so one buffer serves any number of
     launches that do not overlap in time -- one buffer per stream, and per captured graph if graphs may replay concurrently: two launches in
     flight on the same pair would steal each other's items.
     With it, launches of at least two rounds of workgroups run as ONE persistent workgroup per CU that takes (query block, head, batch)
     items from `next`: the hardware deals workgroup ids round-robin over the 8 XCDs whatever their clocks (up to 6 % apart under the power
     cap), which left the faster XCDs idle for ~2.4 % of every fusion-attention launch (profiles/r05_*per_xcd*).  Other launches ignore it. */
  uint32_t* sched_counter;
  /* Operand planes of Q and K (ABI 340, precision "robust"): 0 / 1 = one 16-bit number per element, as described above.  3 = like 2 with the two
     CORRECTION products on the block-scaled fp8 MFMA: a head of a q / K row is [hi fp16 (128 bytes) | e4m3(hi) (64 bytes) | e4m3(lo 2^12) (64 bytes)]
     (f3r_qkv_planes with planes = 3; the same 256 bytes per head) and a score block is q_hi k_hi on fp16 + q_lo8 k_hi8 + q_hi8 k_lo8 on
     v_mfma_scale_f32_32x32x64_f8f6f4 (kernel f3r_attn_asm_qk3f8_f16: 256 matrix-pipe cycles per block instead of 384; the fp8 copies move the
     softmax by ~2e-5, csrc/asm/attn_gen.py corr = "f8").  2 = every head of a q row
     and of a K row holds [hi (64) | lo (64)] fp16 (x = hi + lo to ~22 bits; f3r_qkv_planes writes such rows): ldq / ldk / the batch strides count
     elements of these 128-wide heads (ldq >= n_heads * 128), and the scores are q_hi k_hi + q_lo k_hi + q_hi k_lo with fp32 accumulation -- three
     MFMA products per score block, P V unchanged (V^T, o, the parked state: head_dim 64 layouts).  Only the hand-scheduled kernel
     f3r_attn_asm_qk3_f16 reads this layout (csrc/asm/attn_gen.py AttnGen(qk_planes = 2): 256-query workgroups): fp16, head_dim 64, q_prescaled, and
     the eligibility rules of kernel_sel 0 with tq >= 64 and no minimum number of keys -- anything else is F3R_ERR_UNSUPPORTED, never a fallback.
     These kernels carry the softmax state at ANY batch: sequence z owns rows [z tq, (z + 1) tq) of st_o / st_ml. */
  int32_t qk_planes;
  /* Compute units the persistent (work-stealing) form leaves FREE (ABI 340; 0 = none; needs sched_counter).  The hand-scheduled kernels hold a
     CU completely -- one 488-register wave per SIMD -- so while a launch of one persistent workgroup per CU runs, no other kernel of more than a
     few registers can become resident anywhere on the chip until it ends.  A view-sharded rank launches its local-shard attention next to the
     RCCL kernels (or copy kernels) that move the other ranks' K / V^T: with reserve_cus = r the launch runs cus - r persistent workgroups
     (whenever it has more work items than that), so a kernel that arrives late still finds r CUs -- at the price of r / cus of this launch's rate
     (tools/ubench/exchange_overlap.hip measures both sides: profiles/r06_exchange_under_persistent_attention.json). */
  int32_t reserve_cus;
} f3r_attn_args;
#define F3R_ATTN_ASM_MIN_KEYS 2048

int f3r_attn_fwd(const f3r_attn_args* args, f3r_stream_t stream);
/* which kernel f3r_attn_fwd takes for `args` (a static string; reporting only) */
const char* f3r_attn_kernel_name(const f3r_attn_args* args);

/* ---------------------------------------------------------------------------------------------
 * f3r_upsample2x: bilinear x2, align_corners=True, NHWC lowp -> NHWC lowp.
 * Replaces F.interpolate(scale_factor=2, mode="bilinear", align_corners=True) in
 * FeatureFusionBlock_custom.forward (dpt_block.py:238-243) and the head's Interpolate (:374).
 * The output may be cropped to (out_h, out_w) <= (2h, 2w) (refinenet4 crop, dpt_head.py:69-71).
 * in_lo / out_lo: optional low planes (split precision, see f3r_gemm_args.split): the input is in + in_lo, the output is
 * written as hi = lowp(y), lo = lowp(y - hi).  Either may be NULL.
 */
int f3r_upsample2x(const void* in, const void* in_lo, void* out, void* out_lo, int batch, int h, int w, int C, int out_h, int out_w,
                   int dtype, f3r_stream_t stream);
/* f3r_interp_bilinear: the same kernel for any nominal output size (full_h, full_w) >= (out_h, out_w): F.interpolate(size or
 * scale_factor, mode="bilinear", align_corners=True), src = dst * (in - 1) / (full - 1).  The head's Interpolate(scale_factor =
 * patch_size / 8) (dpt_block.py:374) is x2 for patch 16 and x1.75 for DINOv2's patch 14. */
int f3r_interp_bilinear(const void* in, const void* in_lo, void* out, void* out_lo, int batch, int h, int w, int C, int full_h, int full_w,
                        int out_h, int out_w, int dtype, f3r_stream_t stream);
/* ABI 350: the same with an optional fp8 copy of the output for the next F3R_SPLIT_X3F8 convolution: out_f8 (NULL = none) is
   [batch][out_h][out_w][2 C] bytes, per pixel [C bytes e4m3(y clamped to +-448) | C bytes e4m3((y - float(fp16(y))) 2^12, clamped)] (f3r_gemm_args.out_f8
   describes the same layout).  fp16 only when out_f8 is given. */
int f3r_interp_bilinear_f8(const void* in, const void* in_lo, void* out, void* out_lo, void* out_f8, int batch, int h, int w, int C, int full_h,
                           int full_w, int out_h, int out_w, int dtype, f3r_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * f3r_dpt_final: last 1x1 conv (Cin -> n_out) fused with postprocess.
 * Replaces head[4] Conv2d(last_dim, 3 + has_conf, 1) (dpt_block.py:379-381) + postprocess / reg_dense_depth /
 * reg_dense_conf (heads/postprocess.py:16-64):
 *   xyz,c = W(n_out,Cin) x + b;  d = |xyz|
 *   depth_mode  F3R_DEPTH_EXP    pts = xyz / max(d, 1e-8) * expm1(d)      ('exp', -inf, inf): the released configuration
 *               F3R_DEPTH_LINEAR pts = xyz                                 ('linear', -inf, inf)
 *               F3R_DEPTH_SQUARE pts = xyz / max(d, 1e-8) * d^2            ('square', -inf, inf)
 *               (the reference asserts the depth bounds are infinite, postprocess.py:33-34)
 *   conf_mode   F3R_CONF_EXP     conf = vmin + min(exp(c), vmax - vmin)    ('exp', vmin, vmax)
 *               F3R_CONF_SIGMOID conf = (vmax - vmin) sigmoid(c) + vmin    ('sigmoid', vmin, vmax)
 *   x (+ x_lo, optional low plane): NHWC lowp [npix][Cin];  w: fp32 [n_out][Cin];  b: fp32 [n_out];  n_out = 3 (no confidence:
 *   conf must be NULL) or 4;  pts3d: fp32 [npix][3];  conf: fp32 [npix] or NULL
 */
typedef enum { F3R_DEPTH_EXP = 0, F3R_DEPTH_LINEAR = 1, F3R_DEPTH_SQUARE = 2 } f3r_depth_mode;
typedef enum { F3R_CONF_EXP = 0, F3R_CONF_SIGMOID = 1 } f3r_conf_mode;
int f3r_dpt_final(const void* x, const void* x_lo, const float* w, const float* b, int n_out, float* pts3d, float* conf, int64_t npix,
                  int Cin, int depth_mode, int conf_mode, float conf_vmin, float conf_vmax, int dtype, f3r_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * f3r_cast_f32_to_lp: fp32 -> lowp copy (hooked residual streams 12/18 as DPT inputs, dpt_head.py:54;
 * weight packing).  out_lo (optional): the low plane lowp(x - float(out)) of the split-precision operand format.
 */
int f3r_cast_f32_to_lp(const float* in, void* out, void* out_lo, int64_t n, int dtype, f3r_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * f3r_align_local_to_global: similarity alignment of the local pointmap of every (view, sample) to its global pointmap.
 * Replaces MultiViewDUSt3RLitModule.align_local_pts3d_to_global (fast3r/models/multiview_dust3r_module.py:427-549), i.e. per
 * problem p of npix pixels: thr = torch.quantile(conf, quantile) (:476); mask = conf >= thr & valid (:479-482), falling back
 * to valid alone below 3 points and to the identity below 3 valid points (:495-510); (R, t, s) =
 * roma.rigid_points_registration(local[mask], global[mask], compute_scaling=True) (:509-511; Umeyama); out = s*(local R^T) + t (:514).
 *   conf [n_prob][npix] fp32, pts_local / pts_global / out [n_prob][npix][3] fp32, valid_mask [n_prob][npix] bytes or NULL,
 *   rts [n_prob][13] fp32 = {R row-major (9), t (3), s}, thr_out [n_prob] fp32 or NULL,
 *   workspace: f3r_align_workspace_bytes(n_prob) bytes, 8-byte aligned.
 */
size_t f3r_align_workspace_bytes(int n_prob);
int f3r_align_local_to_global(const float* conf, const float* pts_local, const float* pts_global, const uint8_t* valid_mask,
                              float* out, float* rts, float* thr_out, void* workspace, size_t ws_bytes, int n_prob,
                              int64_t npix, float quantile, f3r_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * f3r_estimate_focal: robust focal length of every view from its pointmap + confidence.
 * Replaces estimate_focal (fast3r/models/multiview_dust3r_module.py:1081-1109): thr = torch.quantile(conf, quantile) (:1089-1093),
 * mask = conf >= thr (:1096), then the "weiszfeld" branch of estimate_focal_knowing_depth_and_confidence_mask
 * (fast3r/dust3r/post_process.py:77-142): closed-form start mean(xy/z . px) / mean(|xy/z|^2) (:121-128), n_iter re-weighted
 * least-squares steps with weights 1 / max(|px - f xy/z|, 1e-8) (:131-136; the reference uses 100), clip to
 * [min_focal, max_focal] x max(H, W) / (2 tan 30deg) (:140-142).  (ppx, ppy) is the principal point; the reference default is (W/2, H/2).
 *   pts3d [n_views][H][W][3] fp32, conf [n_views][H][W] fp32, focal [n_views] fp32, thr_out [n_views] fp32 or NULL,
 *   workspace: f3r_focal_workspace_bytes(n_views, H, W) bytes, 16-byte aligned.
 */
size_t f3r_focal_workspace_bytes(int n_views, int H, int W);
int f3r_estimate_focal(const float* pts3d, const float* conf, float* focal, float* thr_out, void* workspace, size_t ws_bytes,
                       int n_views, int H, int W, float quantile, float ppx, float ppy, int n_iter, float min_focal, float max_focal,
                       f3r_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * f3r_estimate_poses: camera pose (cam-to-world) and, when unknown, focal length of every view from its global pointmap.
 * Replaces the per-view body of MultiViewDUSt3RLitModule.estimate_camera_poses (fast3r/models/multiview_dust3r_module.py:807-869,
 * estimate_cam_pose_one_sample :1038-1078) = fast_pnp (fast3r/dust3r/cloud_opt/init_im_poses.py:300-350): mask = conf > conf_thr
 * (:1045 uses 1.0); fewer than 4 masked points -> failure (:302-303); focal_in[v] > 0 is used as is, otherwise the focal is searched
 * on np.geomspace(max(H,W)/2, 3 max(H,W), n_focals) (:312-316; the reference uses 100) by inlier count at 5 px (:335,:342);
 * result = inverse of the world-to-camera [R|T] (:349-350).  The reference solves with cv2.solvePnPRansac(iterationsCount = niter_PnP,
 * SQPNP); this library uses min(n_iter, 32) sampled closed-form DLT hypotheses + inlier DLT + gated Gauss-Newton (f3r_pnp.hip) -- same
 * contract (n_iter = the RANSAC iteration count), different solver.
 *   pts3d [n_views][H][W][3] fp32 (world frame), conf [n_views][H][W] fp32, focal_in [n_views] fp32 or NULL,
 *   focal_out [n_views] fp32 (NaN on failure), cam_to_world [n_views][4][4] fp32 row-major (identity on failure, :1062-1064),
 *   inliers [n_views] int32 (0 on failure).
 */
int f3r_estimate_poses(const float* pts3d, const float* conf, const float* focal_in, float* focal_out, float* cam_to_world, int* inliers,
                       int n_views, int H, int W, float conf_thr, float ppx, float ppy, int n_focals, int n_iter, f3r_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * f3r_resample_u8 / f3r_imgnorm_u8: the device side of the input pipeline `load_images` (fast3r/dust3r/utils/image.py:76-159).
 * f3r_resample_u8 is ONE pass of PIL.Image.resize for 8-bit RGB (`_resize_pil_image`, :68-74; Pillow's Resample.c): axis 1 =
 * horizontal ([H][W][3] -> [H][out_size][3]), axis 0 = vertical ([H][W][3] -> [out_size][W][3]); bounds [out_size][2] = {first source
 * index, count}, kk [out_size][ksize] = 22-bit fixed-point weights, both computed by the host exactly as Pillow computes them
 * (fast3r_amd/image.py); accumulator 1 << 21, >> 22, saturate: bit-exact with PIL.  f3r_imgnorm_u8 crops (:131-139) and applies
 * `ImgNorm` (:32: ToTensor + Normalize(0.5, 0.5)): [H][W][3] uint8 -> [3][h][w] fp32 = ((u / 255) - 0.5) / 0.5.
 */
int f3r_resample_u8(const uint8_t* in, uint8_t* out, int H, int W, int axis, int out_size, const int32_t* bounds, const int32_t* kk,
                    int ksize, f3r_stream_t stream);
int f3r_imgnorm_u8(const uint8_t* in, float* out, int H, int W, int x0, int y0, int w, int h, f3r_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * LlamaDecoder variant (fast3r/models/fast3r.py:810-968, fast3r/models/components/llama.py): RMSNorm is f3r_layernorm with rms = 1;
 * f3r_silu_mul: SwiGLU gate of FeedForward.forward (llama.py:284), out[r][j] = silu(ab[r][j]) * ab[r][hidden + j], lowp in / out
 * ([rows][2*hidden] -> [rows][hidden]; the two projections w1, w3 are one GEMM with stacked weights);
 * f3r_rows_add_f32: x[r][:] += vec for r < rows -- `x + view0_mask * view0_embed` before every layer (fast3r.py:957-958; the view-0
 * tokens are the first rows of the sequence).
 */
int f3r_silu_mul(const void* ab, void* out, int64_t rows, int hidden, int dtype, f3r_stream_t stream);
int f3r_rows_add_f32(float* x, const float* vec, int64_t rows, int D, f3r_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * precision "exact" -- what `inference(dtype="32")` means in the reference (fast3r/dust3r/inference_multiview.py:41-52: no autocast, fp32
 * everywhere).  Every GEMM / conv of that mode is f3r_gemm with split = F3R_SPLIT_X3 (both operands as hi + lo planes); the attention
 * core (blocks.py:158-169) runs in plain fp32 on q / k / v taken from an fp32 buffer -- a validation mode for scenes of tens of views.
 * f3r_rope2d_f32: RoPE-2D (pos_embed.py:162-183) in place on the q and k parts of qkv[rows][ld] (the first 2 * n_heads * 64 columns);
 *   tables as in f3r_gemm_args.rope_cos / rope_sin ([n_pos][16]), token t of a sequence sits at (t / rope_w, t % rope_w).
 * f3r_attn_f32: o[r][h*64 + d] = sum_j softmax_j(q[r] . k[j] * scale) v[j][d] over the keys of r's sequence (n_seq sequences of seq_len
 *   rows; q, k, v row stride ld floats, head_dim floats per head: 0 or 64, or another multiple of 16 up to 128).  Output as fp32 (o_f32) and / or as lowp hi [+ lo] planes ([rows][ldo]), the
 *   A operand of the X3 projection that follows.
 */
int f3r_rope2d_f32(float* qkv, int64_t rows, int64_t ld, int n_heads, int64_t seq_len, int rope_w, const float* rope_cos,
                   const float* rope_sin, f3r_stream_t stream);
int f3r_attn_f32(const float* q, const float* k, const float* v, int64_t ld, void* o_hi, void* o_lo, float* o_f32, int64_t ldo,
                 int64_t n_seq, int64_t seq_len, int n_heads, float scale, int dtype, int head_dim, f3r_stream_t stream);

/* The general forms (LlamaDecoder and view-sharded models in precision "exact"):
 * f3r_rope_f32: rotary embedding in place on the first n_rot_heads 64-wide column groups of qkv (q heads, then k heads); rope_mode as in
 *   f3r_gemm_args.rope_mode (0: RoPE-2D tables [n_pos][16]; 1: one [32]-angle row per group of rope_w rows, llama.py:96-122).
 * f3r_silu_mul_f32: SwiGLU gate (llama.py:284) on fp32 ab[rows][2*hidden] -> hi + lo planes [rows][hidden].
 * f3r_attn_f32_ex: fp32 attention with separate query / key counts (tq, tk per sequence), grouped-query heads (query head h reads K / V
 *   head h / kv_group: repeat_kv, llama.py:195-198) and an optional causal mask on absolute positions (key j visible to query i iff
 *   k_pos0 + j <= q_pos0 + i), so a rank of a view-sharded model can attend its queries over the gathered keys of all ranks. */
typedef struct f3r_attn_f32_args {
  const float* q;    /* [n_seq * tq][ldq], head h at columns [h * head_dim, (h + 1) * head_dim) */
  const float* k;    /* [n_seq * tk][ldkv], K / V head g at columns [g * head_dim, ...) */
  const float* v;
  int64_t ldq, ldkv;
  void* o_hi;        /* lowp planes [n_seq * tq][ldo] (o_lo optional), and / or */
  void* o_lo;
  float* o_f32;      /* fp32 [n_seq * tq][ldo] */
  int64_t ldo;
  int64_t n_seq, tq, tk;
  int64_t q_pos0, k_pos0; /* causal only */
  int32_t n_heads;   /* query heads */
  int32_t kv_group;  /* query heads per K / V head (0 / 1 = plain multi-head) */
  int32_t causal;
  int32_t dtype;     /* f3r_dtype of the lowp planes */
  int32_t head_dim;  /* 0 = 64, or a multiple of 16 up to 128 */
  float scale;
} f3r_attn_f32_args;
int f3r_rope_f32(float* qkv, int64_t rows, int64_t ld, int n_rot_heads, int64_t seq_len, int rope_w, int rope_mode, const float* rope_cos,
                 const float* rope_sin, f3r_stream_t stream);
int f3r_silu_mul_f32(const float* ab, void* out_hi, void* out_lo, int64_t rows, int hidden, int dtype, f3r_stream_t stream);
int f3r_attn_f32_ex(const f3r_attn_f32_args* args, f3r_stream_t stream);
/* The same attention on the matrix pipe (f3r_exact_mfma.hip; ABI 320): every operand as an exact sum of two 16-bit planes (~22 significand bits),
 * every product as three MFMAs with fp32 accumulation, softmax in fp32 -- the FMA-pipe kernel's numbers to ~1e-6 at 10x its speed, for the
 * scenes where that kernel takes minutes (it stays the reference implementation of the mode; tests compare the two).  head_dim 64, no causal
 * mask: f3r_attn_f32_mfma_workspace returns 0 for anything else, and the bytes of caller-owned scratch (the planes of q, k and V^T) otherwise. */
int64_t f3r_attn_f32_mfma_workspace(const f3r_attn_f32_args* args);
int f3r_attn_f32_mfma(const f3r_attn_f32_args* args, void* workspace, int64_t workspace_bytes, f3r_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * precision "robust" (ABI 340) -- between "high" (every operand one fp16 number, weights hi + lo) and "exact": every GEMM / conv as in "exact"
 * (F3R_SPLIT_X3), the attention core (blocks.py:158-190) with Q and K as hi + lo planes (f3r_attn_args.qk_planes = 2: three products per score
 * block) and P, V single fp16 -- the cheapest operand set that keeps a noise-amplifying checkpoint within 1e-3 of the fp32 path
 * (oracle/precision_study.py --study robust_vitl).  The two passes around that attention launch:
 * f3r_qkv_planes: qkv fp32 [n_seq * seq_len][ld] = q | k | v column blocks of n_heads * 64 | kv_heads * 64 | kv_heads * 64 (the output of the QKV
 *   projection; rotary embedding already applied) -> q_planes [rows][n_heads][hi 64 | lo 64] (values multiplied by q_scale = softmax scale x
 *   log2(e) BEFORE the split), k_planes [rows][kv_heads][hi 64 | lo 64], vt [n_seq][kv_heads * 64][ldvt] one plane (key columns >= seq_len zero).
 * f3r_attn_state_finish: the state an attention launch parked (f3r_attn_args.state_out: st_o fp32 [rows][n_heads * head_dim] un-normalised,
 *   st_ml fp32 [rows][n_heads][4] = {m, l of lane half 0, of lane half 1, -}) -> O / (l0 + l1) as hi + lo planes [rows][ldo] (the A operand of the
 *   X3 output projection; o_lo may be NULL) and / or fp32 (o_f32).
 */
int f3r_qkv_planes(const float* qkv, int64_t ld, int64_t n_seq, int64_t seq_len, int n_heads, int kv_heads, float q_scale, void* q_planes, void* k_planes,
                   void* vt, int64_t ldvt, int dtype, int planes /* 2 or 3: the row layout of f3r_attn_args.qk_planes */, f3r_stream_t stream);
int f3r_attn_state_finish(const float* st_o, const float* st_ml, int64_t rows, int n_heads, int head_dim, void* o_hi, void* o_lo, float* o_f32,
                          int64_t ldo, int dtype, f3r_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* F3R_H_ */
