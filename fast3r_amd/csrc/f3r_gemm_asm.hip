// Host side of the hand-scheduled GEMM kernels: the code object assembled from csrc/asm/gemm_gen.py is embedded in the library
// (obj/f3r_gemm_asm_blob.cpp, written by build.sh), loaded once per device with hipModuleLoadData and launched with hipModuleLaunchKernel
// on the caller's stream.  f3r_gemm (f3r_gemm.hip) decides per call whether a launch goes here (f3r_gemm_args.kernel_sel, include/f3r.h).
#include <cstddef>
#include <cstring>
#include <map>
#include <mutex>

#include "f3r_common.h"

extern "C" const unsigned char f3r_gemm_asm_hsaco[];
extern "C" const unsigned int f3r_gemm_asm_hsaco_len;

namespace {

// kernel argument block: the ARG_* offsets of csrc/asm/gemm_gen.py
struct f3r_gemm_asm_args {
  const void* A;
  const void* W;
  const float* bias;
  const float* res;
  void* out;
  uint32_t lda_b, ldw_b, ldr_b, ldo_b;  // row strides in bytes
  uint32_t nk, nk1;                     // K-tiles over all K segments / per segment
  uint32_t xq, xr, pg, pg_magic, gm_shift, act, grid, n_wg;  // tile map (gemm_gen.pack_args), activation, persistent grid: workgroup b
                                                             // computes output tiles b, b + grid, b + 2 grid, ...
  int64_t seg_stride;                   // ARG_SEG: output segments along n (bytes from one segment's buffer to the next)
  uint32_t tps, tps_magic;              // n tiles per segment, ceil(2^32 / tps)
  float scale;                          // ACT_SCALE: segment 0 is multiplied by it
  uint32_t flags, nk1_w, pad;           // FLAG_BIAS_ON_M = 1; wrap period of the W stream
  // ---- ARG_F8: read by the f8 kernels only (their kernarg segment is 144 bytes, the others' 128)
  const uint32_t* w_scale;              // E8M0 scale words of the weight rows' fp8 plane
  uint32_t nk8, out8_off;               // fp8 K-tiles ([256][128 k]) at the end of every output tile's K loop; byte offset of the output's fp8 copy in its row (0 = none)
  // ---- ARG_ROPE: read by the lowp-role kernels only (kernarg segment 176 bytes), ACT_ROPE
  const float* rope_cos;                // [n_pos][16]
  const float* rope_sin;
  uint32_t seq_len, seq_magic, rope_w, rope_w_magic;   // token m of the launch sits at position m % seq_len = (y, x) = (pos / rope_w, pos % rope_w)
};
static_assert(sizeof(f3r_gemm_asm_args) == 176 && offsetof(f3r_gemm_asm_args, w_scale) == 128 && offsetof(f3r_gemm_asm_args, rope_cos) == 144 && offsetof(f3r_gemm_asm_args, lda_b) == 40 && offsetof(f3r_gemm_asm_args, xq) == 64 &&
              offsetof(f3r_gemm_asm_args, seg_stride) == 96 && offsetof(f3r_gemm_asm_args, nk1_w) == 120, "must match ARG_* of gemm_gen.py");
enum { ACT_SCALE = 3, ACT_ROPE = 4, FLAG_BIAS_ON_M = 1, FLAG_SKEW = 2 };

enum { ROLE_F32 = 0, ROLE_LP = 1 };
struct DevKernels {
  bool tried = false;
  hipModule_t mod = nullptr;
  hipFunction_t fn[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [role][F3R_F16 | F3R_BF16]
  hipFunction_t fn8[2] = {nullptr, nullptr};                          // [role]: the fp16 kernels with the low plane in fp8
};
std::map<int, DevKernels> g_dev;  // one code-object handle per device (see f3r_attn_asm.hip)
std::mutex g_mu;

hipFunction_t get_fn(int role, int dtype, bool f8 = false) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dtype < 0 || dtype > 1 || (f8 && dtype != F3R_F16)) return nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  DevKernels& d = g_dev[dev];
  if (!d.tried) {
    d.tried = true;
    if (hipModuleLoadData(&d.mod, f3r_gemm_asm_hsaco) == hipSuccess) {
      static const char* names[2][2] = {{"f3r_gemm_asm_f32_f16", "f3r_gemm_asm_f32_bf16"}, {"f3r_gemm_asm_lp_f16", "f3r_gemm_asm_lp_bf16"}};
      for (int r = 0; r < 2; ++r)
        for (int t = 0; t < 2; ++t)
          if (hipModuleGetFunction(&d.fn[r][t], d.mod, names[r][t]) != hipSuccess) d.fn[r][t] = nullptr;
      static const char* names8[2] = {"f3r_gemm_asm_f328_f16", "f3r_gemm_asm_lp8_f16"};
      for (int r = 0; r < 2; ++r)
        if (hipModuleGetFunction(&d.fn8[r], d.mod, names8[r]) != hipSuccess) d.fn8[r] = nullptr;
    }
    (void)hipGetLastError();
  }
  return f8 ? d.fn8[role] : d.fn[role][dtype];
}

int role_of(const f3r_gemm_args& a) { return a.out_f32 ? ROLE_F32 : ROLE_LP; }

int num_cus() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
  return cus / 8 * 8;
}

}  // namespace

// The kernel's tile map divides a tile position by pg = gm x (n tiles) and an n tile index by tps (n tiles per output segment) through
// floor(x * ceil(2^32 / d) / 2^32), which is exact only while x * d < 2^32 (gemm_gen.pack_args): both products must stay below that.
static bool tile_map_exact(int64_t ntm, int64_t ntn, int64_t tps) {
  int gsh = 3;
  while (ntm % (1ll << gsh)) --gsh;
  const int64_t n_wg = ntm * ntn, pg = (1ll << gsh) * ntn;
  return n_wg * pg < (1ll << 32) && ntn * (tps > 0 ? tps : 1) < (1ll << 32);
}

// Is a launch of `tiles` 256 x 256 tiles worth the persistent one-workgroup-per-CU grid?  At least one full round, and the last round of
// the tile walk at least 80 % full (320 tiles on 256 CUs would leave three quarters of the chip idle for half of the launch; the 8-wave
// kernel scores its three tile forms for such launches, f3r_gemm256_impl.h tile_score).
bool f3r_gemm_asm_preferred(int64_t tiles) {
  const int64_t cus = num_cus();
  if (tiles < cus) return false;
  const int64_t rounds = (tiles + cus - 1) / cus;
  return tiles * 5 >= rounds * cus * 4;
}

// Can the hand-scheduled kernel take this (already validated) launch?  *why names the first obstacle.
bool f3r_gemm_asm_eligible(const f3r_gemm_args& a, const char** why) {
  *why = "";
  if (a.a_mode != F3R_A_PLAIN) { *why = "not a plain GEMM operand (convolution)"; return false; }
  if (a.epi != F3R_EPI_GENERIC) { *why = "QKV / ConvT epilogue"; return false; }
  if (a.split != F3R_SPLIT_NONE && a.split != F3R_SPLIT_W2) { *why = "X3 split"; return false; }
  if (a.M <= 0 || a.M % 256 != 0 || a.N % 256 != 0) { *why = "M or N not a multiple of 256"; return false; }
  const int Kpad1 = a.split ? a.Kpad / 2 : a.Kpad;
  if (a.K != Kpad1 || Kpad1 % 64 != 0) { *why = "K is not the padded depth (a K tail cannot be zero-filled by LDS-DMA)"; return false; }
  if ((a.Kpad / 64) < 4) { *why = "fewer than 4 K-tiles (the operand streams run three K-tiles ahead)"; return false; }
  if (a.rowadd || a.res_lp || a.res_lp2 || a.out_lp_lo || a.out_relu || a.out_relu_lo || a.out_f8 || a.out_relu_f8 || a.fin_w) { *why = "additive rows / lowp residuals / second outputs"; return false; }
  if (a.out_f32 && a.out_lp) { *why = "both an fp32 and a lowp output"; return false; }
  if (a.out_f32) {
    if (a.act != F3R_ACT_NONE) { *why = "activation on the fp32 role"; return false; }
    if ((a.ldo_f32 * 4) % 16 != 0 || (a.res_f32 && (a.ldr_f32 * 4) % 16 != 0)) { *why = "fp32 row strides not multiples of 16 bytes"; return false; }
    if ((int64_t)256 * a.ldo_f32 * 4 >= (1ll << 32) || (a.res_f32 && (int64_t)256 * a.ldr_f32 * 4 >= (1ll << 32))) { *why = "fp32 row strides too large"; return false; }
  } else {
    if (a.res_f32) { *why = "fp32 residual with a lowp output"; return false; }
    if ((((uintptr_t)a.out_lp) & 15) != 0 || (a.ldo_lp * 2) % 16 != 0) { *why = "lowp output not 16-byte aligned"; return false; }
    if ((int64_t)256 * a.ldo_lp * 2 >= (1ll << 32)) { *why = "lowp row stride too large"; return false; }
  }
  if ((int64_t)256 * a.lda * 2 >= (1ll << 32) || (int64_t)256 * a.Kpad * 2 >= (1ll << 32)) { *why = "operand row strides too large for 32-bit lane offsets"; return false; }
  const int64_t tiles = (a.M / 256) * (a.N / 256);
  if (tiles >= (1ll << 24)) { *why = "grid too large"; return false; }
  if (!tile_map_exact(a.M / 256, a.N / 256, a.N / 256)) { *why = "tile map: position x period reaches 2^32 (the kernel divides by multiplying with ceil(2^32 / period))"; return false; }
  if (get_fn(role_of(a), a.dtype) == nullptr) { *why = "the embedded code object could not be loaded on this device"; return false; }
  return true;
}

namespace {

// fills the tile map / persistent grid and launches one kernel: M x N outputs in 256 x 256 tiles
// role / f8 select the kernarg segment the kernel declares: lowp role 176 bytes (.. ARG_ROPE), fp32 role 128, or 144 with the fp8 low plane
int launch_tiles(hipFunction_t fn, f3r_gemm_asm_args& k, int64_t M, int64_t N, hipStream_t stream, bool skew, int role) {
  const uint32_t ntm = (uint32_t)(M / 256), ntn = (uint32_t)(N / 256);
  const uint32_t n_wg = ntm * ntn;
  // tile map (gemm_gen.pack_args): XCD-contiguous runs, groups of gm m-tiles x all n-tiles, gm = the largest power of two <= 8 dividing ntm
  uint32_t gsh = 3;
  while (ntm % (1u << gsh)) --gsh;
  k.xq = n_wg / 8;
  k.xr = n_wg % 8;
  k.pg = (1u << gsh) * ntn;
  k.pg_magic = k.pg > 1 ? (uint32_t)(((1ull << 32) + k.pg - 1) / k.pg) : 0;
  k.gm_shift = gsh;
  if (k.tps == 0) k.tps = ntn;
  k.tps_magic = k.tps > 1 ? (uint32_t)(((1ull << 32) + k.tps - 1) / k.tps) : 0;
  // persistent grid: one workgroup per CU (160 KiB of LDS, 512 registers per lane: one resident workgroup), a multiple of 8 so that a
  // workgroup's tiles stay on its XCD's contiguous run
  const int cus = num_cus();
  k.n_wg = n_wg;
  k.grid = n_wg < (uint32_t)cus ? n_wg : (uint32_t)cus;
  // kernel_sel 9 (measurement): start the persistent workgroups up to 3/4 of a tile period apart, so that one quarter's epilogue traffic overlaps
  // the others' K loops instead of all 256 CUs writing out at once.  MEASURED SLOWER (profiles/r05_gemm_w2_vs_w2f8_and_start_skew.jsonl: fc2 -6 %,
  // fc1 -0.7 %, proj +1 %): workgroups that run in lock step share their operand panels through the XCD's L2 in time; off by default.
  if (skew && n_wg >= 2 * k.grid) k.flags |= FLAG_SKEW;
  size_t size = role == ROLE_LP ? sizeof(k) : (k.w_scale ? offsetof(f3r_gemm_asm_args, rope_cos) : offsetof(f3r_gemm_asm_args, w_scale));
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  hipError_t e = hipModuleLaunchKernel(fn, k.grid, 1, 1, 256, 1, 1, 0, stream, nullptr, config);
  if (e != hipSuccess) {
    f3r_set_error("f3r_gemm: hipModuleLaunchKernel failed: %s", hipGetErrorString(e));
    return F3R_ERR_LAUNCH;
  }
  return f3r_check_launch("f3r_gemm(asm)");
}

}  // namespace

int f3r_gemm_asm_launch(const f3r_gemm_args& a, hipStream_t stream) {
  const int role = role_of(a);
  hipFunction_t fn = get_fn(role, a.dtype);
  if (!fn) {
    f3r_set_error("f3r_gemm: the embedded hand-scheduled kernel could not be loaded on this device");
    return F3R_ERR_LAUNCH;
  }
  f3r_gemm_asm_args k;
  memset(&k, 0, sizeof(k));
  const int planes = a.split ? 2 : 1;
  const int Kpad1 = a.Kpad / planes;
  k.A = a.A;
  k.W = a.W;
  k.bias = a.bias;
  k.res = role == ROLE_F32 ? a.res_f32 : nullptr;
  k.out = role == ROLE_F32 ? (void*)a.out_f32 : a.out_lp;
  k.lda_b = (uint32_t)(a.lda * 2);
  k.ldw_b = (uint32_t)((int64_t)a.Kpad * 2);
  k.ldr_b = (uint32_t)(a.ldr_f32 * 4);
  k.ldo_b = role == ROLE_F32 ? (uint32_t)(a.ldo_f32 * 4) : (uint32_t)(a.ldo_lp * 2);
  k.nk1 = (uint32_t)(Kpad1 / 64);
  k.nk = k.nk1 * (uint32_t)planes;
  k.nk1_w = k.nk;  // the weight row holds its planes back to back: that stream never wraps
  k.act = (uint32_t)a.act;
  k.scale = 1.0f;
  return launch_tiles(fn, k, a.M, a.N, stream, a.kernel_sel == 9, role);
}

// ---- F3R_SPLIT_W2F8: rows [K fp16 | K fp8] on both operands, the K loop runs on from K / 64 fp16 K-tiles into K / 128 fp8 ones
bool f3r_gemm_asm_f8_eligible(const f3r_gemm_args& a, const char** why) {
  *why = "";
  if (a.dtype != F3R_F16) { *why = "the fp8 low plane corrects fp16 planes only"; return false; }
  if (a.a_mode != F3R_A_PLAIN || a.epi != F3R_EPI_GENERIC) { *why = "not a plain GEMM with the generic epilogue"; return false; }
  if (a.M <= 0 || a.M % 256 != 0 || a.N % 256 != 0) { *why = "M or N not a multiple of 256"; return false; }
  if (a.K % 128 != 0 || a.K / 64 + a.K / 128 < 4) { *why = "K not a multiple of 128, or fewer than 4 K-tiles"; return false; }
  if (a.rowadd || a.res_lp || a.res_lp2 || a.out_lp_lo || a.out_relu || a.out_relu_lo || a.A_lo) { *why = "additive rows / lowp residuals / second outputs / A_lo"; return false; }
  if (a.out_f32 && a.out_lp) { *why = "both an fp32 and a lowp output"; return false; }
  if (a.out_f32) {
    if (a.act != F3R_ACT_NONE) { *why = "activation on the fp32 role"; return false; }
    if ((a.ldo_f32 * 4) % 16 != 0 || (a.res_f32 && (a.ldr_f32 * 4) % 16 != 0)) { *why = "fp32 row strides not multiples of 16 bytes"; return false; }
    if ((int64_t)256 * a.ldo_f32 * 4 >= (1ll << 32) || (a.res_f32 && (int64_t)256 * a.ldr_f32 * 4 >= (1ll << 32))) { *why = "fp32 row strides too large"; return false; }
  } else {
    if (a.res_f32) { *why = "fp32 residual with a lowp output"; return false; }
    if ((((uintptr_t)a.out_lp) & 15) != 0 || (a.ldo_lp * 2) % 16 != 0) { *why = "lowp output not 16-byte aligned"; return false; }
    if ((int64_t)256 * a.ldo_lp * 2 >= (1ll << 32)) { *why = "lowp row stride too large"; return false; }
    if (a.out_lp_f8 && (a.act != F3R_ACT_GELU || a.ldo_lp * 2 < (int64_t)a.N * 3)) { *why = "out_lp_f8 needs the GELU epilogue and rows of >= 3 N / 2 elements"; return false; }
  }
  if (a.out_lp_f8 && a.out_f32) { *why = "out_lp_f8 with an fp32 output"; return false; }
  if ((int64_t)256 * a.lda * 2 >= (1ll << 32) || (int64_t)256 * a.K * 3 >= (1ll << 32)) { *why = "operand row strides too large for 32-bit lane offsets"; return false; }
  if ((a.M / 256) * (int64_t)(a.N / 256) >= (1ll << 24) || !tile_map_exact(a.M / 256, a.N / 256, a.N / 256)) { *why = "grid too large"; return false; }
  if (get_fn(role_of(a), a.dtype, true) == nullptr) { *why = "the embedded code object could not be loaded on this device"; return false; }
  return true;
}

int f3r_gemm_asm_f8_launch(const f3r_gemm_args& a, hipStream_t stream) {
  const int role = role_of(a);
  hipFunction_t fn = get_fn(role, a.dtype, true);
  if (!fn) {
    f3r_set_error("f3r_gemm: the embedded hand-scheduled kernel could not be loaded on this device");
    return F3R_ERR_LAUNCH;
  }
  f3r_gemm_asm_args k;
  memset(&k, 0, sizeof(k));
  k.A = a.A;
  k.W = a.W;
  k.bias = a.bias;
  k.res = role == ROLE_F32 ? a.res_f32 : nullptr;
  k.out = role == ROLE_F32 ? (void*)a.out_f32 : a.out_lp;
  k.lda_b = (uint32_t)(a.lda * 2);
  k.ldw_b = (uint32_t)((int64_t)a.K * 3);
  k.ldr_b = (uint32_t)(a.ldr_f32 * 4);
  k.ldo_b = role == ROLE_F32 ? (uint32_t)(a.ldo_f32 * 4) : (uint32_t)(a.ldo_lp * 2);
  k.nk8 = (uint32_t)(a.K / 128);
  k.nk = (uint32_t)(a.K / 64) + k.nk8;
  k.nk1 = k.nk;     // neither stream wraps: both rows hold the fp8 plane behind the fp16 one
  k.nk1_w = k.nk;
  k.act = (uint32_t)a.act;
  k.scale = 1.0f;
  k.w_scale = a.w_scale;
  k.out8_off = a.out_lp_f8 ? (uint32_t)a.N * 2u : 0u;
  return launch_tiles(fn, k, a.M, a.N, stream, a.kernel_sel == 9, role);
}

// ---- the QKV projection without rotary embedding (the fusion decoder: blocks.py:138-143 with rope = None) as two launches of the lowp
// role: q | k columns into their two buffers (output segments; q scaled by q_scale), then V^T = W_v X^T with the operand roles swapped
// (the kernel's A operand = the v rows of the fused weight incl. their lo plane, its W operand = the activations; bias by output row; one
// output segment per sequence).  The activations are read twice; the weights once.
bool f3r_gemm_asm_qkv_eligible(const f3r_gemm_args& a, const char** why) {
  *why = "";
  if (a.epi != F3R_EPI_QKV || a.a_mode != F3R_A_PLAIN) { *why = "not a QKV launch"; return false; }
  if (a.rope_cos) {  // round 5: RoPE-2D of the CroCo encoder fused into the q | k launch's epilogue (ACT_ROPE); the per-group form of the LlamaDecoder is not
    if (a.rope_mode != 0 || a.rope_w <= 0) { *why = "rotary embedding per row group (rope_mode 1)"; return false; }
    if (a.M * (int64_t)a.seq_len >= (1ll << 32) || (int64_t)a.seq_len * a.rope_w >= (1ll << 32)) { *why = "token index x sequence length reaches 2^32 (magic-number division)"; return false; }
  }
  if (a.split != F3R_SPLIT_NONE && a.split != F3R_SPLIT_W2 && a.split != F3R_SPLIT_W2F8) { *why = "X3 split"; return false; }
  const bool f8 = a.split == F3R_SPLIT_W2F8;
  if (f8 && (a.dtype != F3R_F16 || !a.w_scale || !a.W_aux || a.K % 128 != 0 || a.Kpad != a.K)) { *why = "W2F8 needs fp16, w_scale, W_aux, K = Kpad a multiple of 128"; return false; }
  const int Dq = a.qkv_dq ? a.qkv_dq : a.N / 3;
  const int Dkv = (a.N - Dq) / 2;
  if (Dq != Dkv) { *why = "grouped-query widths (q and k segments differ)"; return false; }
  if (Dq % 256 != 0 || a.M <= 0 || a.M % 256 != 0 || a.seq_len % 256 != 0) { *why = "D, M or seq_len not a multiple of 256"; return false; }
  const int Kpad1 = (a.split && !f8) ? a.Kpad / 2 : a.Kpad;
  if (a.K != Kpad1 || Kpad1 % 64 != 0 || a.Kpad / 64 < 4) { *why = "K tail or fewer than 4 K-tiles"; return false; }
  if ((((uintptr_t)a.q) & 15) || (((uintptr_t)a.k) & 15) || (((uintptr_t)a.vt) & 15) || (a.ldvt * 2) % 16 != 0) { *why = "outputs not 16-byte aligned"; return false; }
  if ((int64_t)256 * a.lda * 2 >= (1ll << 32) || (int64_t)256 * a.Kpad * 3 >= (1ll << 32) || (int64_t)256 * a.ldvt * 2 >= (1ll << 32)) { *why = "row strides too large"; return false; }
  if ((a.M / 256) * (int64_t)(a.N / 256) >= (1ll << 24)) { *why = "grid too large"; return false; }
  {
    const int64_t Dq256 = (a.qkv_dq ? a.qkv_dq : a.N / 3) / 256;
    // launch 1: M/256 x 2D/256 tiles in segments of D/256; launch 2 (V^T, operands swapped): D/256 x M/256 tiles in segments of seq_len/256
    if (!tile_map_exact(a.M / 256, 2 * Dq256, Dq256) || !tile_map_exact(Dq256, a.M / 256, a.seq_len / 256)) {
      *why = "tile map: position x period reaches 2^32";
      return false;
    }
  }
  if (get_fn(ROLE_LP, a.dtype) == nullptr || (f8 && get_fn(ROLE_LP, a.dtype, true) == nullptr)) { *why = "the embedded code object could not be loaded on this device"; return false; }
  return true;
}

int f3r_gemm_asm_qkv_launch(const f3r_gemm_args& a, hipStream_t stream) {
  hipFunction_t fn = get_fn(ROLE_LP, a.dtype);
  if (!fn) {
    f3r_set_error("f3r_gemm: the embedded hand-scheduled kernel could not be loaded on this device");
    return F3R_ERR_LAUNCH;
  }
  const bool f8 = a.split == F3R_SPLIT_W2F8;
  const int planes = (a.split && !f8) ? 2 : 1;
  const int Kpad1 = a.Kpad / planes;
  const int D = a.qkv_dq ? a.qkv_dq : a.N / 3;
  f3r_gemm_asm_args k;
  memset(&k, 0, sizeof(k));
  // launch 1: q | k
  k.A = a.A;
  k.W = a.W;
  k.bias = a.bias;
  k.out = a.q;
  k.lda_b = (uint32_t)(a.lda * 2);
  k.ldo_b = (uint32_t)(D * 2);
  if (f8) {  // rows [K fp16 | K fp8] on both operands: K / 64 fp16 K-tiles, then K / 128 fp8 ones; neither stream wraps
    hipFunction_t fn8 = get_fn(ROLE_LP, a.dtype, true);
    if (!fn8) {
      f3r_set_error("f3r_gemm: the embedded hand-scheduled kernel could not be loaded on this device");
      return F3R_ERR_LAUNCH;
    }
    k.ldw_b = (uint32_t)((int64_t)a.K * 3);
    k.nk8 = (uint32_t)(a.K / 128);
    k.nk = (uint32_t)(a.K / 64) + k.nk8;
    k.nk1 = k.nk1_w = k.nk;
    k.w_scale = a.w_scale;
  } else {
    k.ldw_b = (uint32_t)((int64_t)a.Kpad * 2);
    k.nk1 = (uint32_t)(Kpad1 / 64);
    k.nk = k.nk1 * (uint32_t)planes;
    k.nk1_w = k.nk;
  }
  k.act = a.rope_cos ? ACT_ROPE : ACT_SCALE;
  k.scale = a.q_scale != 0.f ? a.q_scale : 1.0f;
  if (a.rope_cos) {
    k.rope_cos = a.rope_cos;
    k.rope_sin = a.rope_sin;
    k.seq_len = (uint32_t)a.seq_len;
    k.seq_magic = a.seq_len > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)a.seq_len - 1) / (uint64_t)a.seq_len) : 0;
    k.rope_w = (uint32_t)a.rope_w;
    k.rope_w_magic = a.rope_w > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)a.rope_w - 1) / (uint64_t)a.rope_w) : 0;
  }
  k.seg_stride = (int64_t)((const char*)a.k - (const char*)a.q);
  k.tps = (uint32_t)(D / 256);
  int rc = launch_tiles(f8 ? get_fn(ROLE_LP, a.dtype, true) : fn, k, a.M, 2 * (int64_t)D, stream, a.kernel_sel == 9, ROLE_LP);
  if (rc != F3R_OK) return rc;
  // launch 2: V^T[seq][d][t] = W_v X^T (W2F8: the v rows as two fp16 planes from W_aux, the activations' fp16 part)
  f3r_gemm_asm_args v;
  memset(&v, 0, sizeof(v));
  const int64_t kp_v = f8 ? (int64_t)2 * a.K : a.Kpad;   // elements per v weight row (both planes)
  v.A = f8 ? a.W_aux : (const void*)((const char*)a.W + (int64_t)2 * D * a.Kpad * 2);
  v.W = a.A;
  v.bias = a.bias ? a.bias + 2 * D : nullptr;
  v.out = a.vt;
  v.lda_b = (uint32_t)(kp_v * 2);
  v.ldw_b = (uint32_t)(a.lda * 2);
  v.ldo_b = (uint32_t)(a.ldvt * 2);
  v.nk = f8 ? (uint32_t)(2 * (a.K / 64)) : k.nk;
  v.nk1 = v.nk;      // the weight planes follow each other in a row: that stream runs on
  v.nk1_w = f8 ? (uint32_t)(a.K / 64) : k.nk1;   // the activations wrap per K segment
  v.act = 0;
  v.scale = 1.0f;
  v.flags = FLAG_BIAS_ON_M;
  v.seg_stride = (int64_t)D * a.ldvt * 2;
  v.tps = (uint32_t)(a.seq_len / 256);
  return launch_tiles(fn, v, D, a.M, stream, a.kernel_sel == 9, ROLE_LP);
}
