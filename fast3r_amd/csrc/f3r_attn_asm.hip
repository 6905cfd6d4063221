// Host side of the hand-scheduled attention kernel: the code object assembled from csrc/asm/attn_gen.py is embedded in the library
// (obj/f3r_attn_asm_blob.cpp, written by build.sh), loaded once per device with hipModuleLoadData and launched with
// hipModuleLaunchKernel on the caller's stream (capturable in a hipGraph like any other launch).  f3r_attn_fwd (f3r_attn.hip)
// decides per call whether a launch goes here (f3r_attn_args.kernel_sel, include/f3r.h).
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "f3r_common.h"

extern "C" const unsigned char f3r_attn_asm_hsaco[];
extern "C" const unsigned int f3r_attn_asm_hsaco_len;

namespace {

// kernel argument block: the ARG_* offsets of csrc/asm/attn_gen.py
struct f3r_attn_asm_seg {
  const void* k;
  const void* vt;
  uint32_t tiles, pad;
};
struct f3r_attn_asm_args {
  const void* q;
  void* o;
  uint32_t ldq_b, ldk_b, ldvt_b, ldo_b;  // row strides in bytes
  uint32_t n_tiles, n_seg;               // 64-key tiles over all segments, number of (non-empty) segments
  uint64_t q_bs, o_bs;                   // batch strides in bytes
  uint32_t kv_shift, flags;              // bit 0: state_in, bit 1: state_out
  float* st_o;
  float* st_ml;
  uint64_t k_bs, vt_bs;
  uint32_t st_o_ld_b, st_ml_ld_b;
  uint32_t tq, pad;                      // query rows (the last workgroup may be partial: ARG_TQ)
  f3r_attn_asm_seg seg[8];
  uint32_t* dbg;                         // ARG_DBG: optional counters (f3r_attn_args.dbg_counters)
  uint32_t* sched;                       // ARG_SCHED: {next, done} of the work-stealing form, or NULL = one work item per workgroup id
  uint32_t n_work, nx, nxy, nx_magic, nxy_magic, grid;   // items, q blocks, q blocks x heads, ceil(2^32 / nx), ceil(2^32 / nxy), workgroups launched
};
static_assert(sizeof(f3r_attn_asm_args) == 344 && offsetof(f3r_attn_asm_args, seg) == 112 && offsetof(f3r_attn_asm_args, dbg) == 304 &&
              offsetof(f3r_attn_asm_args, sched) == 312 && offsetof(f3r_attn_asm_args, n_work) == 320,
              "must match ARG_SIZE / ARG_SEG / ARG_DBG / ARG_SCHED of attn_gen.py");

// The code object is loaded once per DEVICE (hipModule handles are per device context): the only process-wide state of the library
// besides the per-thread error string.  Any device index: the table grows on demand.
// The generated kernels: head_dim 64 (512 queries per workgroup), 80 and 128 (256 queries per workgroup); HEAD_DIMS of attn_gen.py
constexpr int kNumHd = 3;
constexpr int kHd[kNumHd] = {64, 80, 128};
int hd_index(int hd) {
  for (int i = 0; i < kNumHd; ++i)
    if (kHd[i] == hd) return i;
  return -1;
}
int wave_rows(int hd, int qk_planes = 1) { return (hd == 64 && qk_planes < 2) ? 128 : 64; }  // query rows of a wave (32 x the query blocks per wave)

struct DevKernels {
  bool tried = false;
  hipModule_t mod = nullptr;
  hipFunction_t fn[kNumHd][2] = {};  // [head_dim index][F3R_F16, F3R_BF16]
  hipFunction_t fn_qk3 = nullptr;    // head_dim 64, fp16, Q and K as hi + lo planes (f3r_attn_args.qk_planes = 2)
  hipFunction_t fn_qk3f8 = nullptr;  // ... with the correction products on the fp8 MFMA (qk_planes = 3)
  hipFunction_t fn_q256[2] = {};     // head_dim 64 with 256-query work items (two query blocks per wave): small launches (f3r_attn_asm_use_q256)
};
std::map<int, DevKernels> g_dev;
std::mutex g_mu;

hipFunction_t get_fn(int dtype, int hd, int qk_planes = 1) {
  int dev = 0;
  const int hi = hd_index(hd);
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dtype < 0 || dtype > 1 || hi < 0) return nullptr;
  if (qk_planes >= 2 && (hd != 64 || dtype != F3R_F16)) return nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  DevKernels& d = g_dev[dev];
  if (!d.tried) {
    d.tried = true;
    if (hipModuleLoadData(&d.mod, f3r_attn_asm_hsaco) == hipSuccess) {
      for (int i = 0; i < kNumHd; ++i)
        for (int t = 0; t < 2; ++t) {
          char name[48];
          if (kHd[i] == 64)
            snprintf(name, sizeof(name), "f3r_attn_asm_%s", t == F3R_F16 ? "f16" : "bf16");
          else
            snprintf(name, sizeof(name), "f3r_attn_asm_d%d_%s", kHd[i], t == F3R_F16 ? "f16" : "bf16");
          if (hipModuleGetFunction(&d.fn[i][t], d.mod, name) != hipSuccess) d.fn[i][t] = nullptr;
        }
      if (hipModuleGetFunction(&d.fn_qk3, d.mod, "f3r_attn_asm_qk3_f16") != hipSuccess) d.fn_qk3 = nullptr;
      if (hipModuleGetFunction(&d.fn_qk3f8, d.mod, "f3r_attn_asm_qk3f8_f16") != hipSuccess) d.fn_qk3f8 = nullptr;
      if (hipModuleGetFunction(&d.fn_q256[F3R_F16], d.mod, "f3r_attn_asm_q256_f16") != hipSuccess) d.fn_q256[F3R_F16] = nullptr;
      if (hipModuleGetFunction(&d.fn_q256[F3R_BF16], d.mod, "f3r_attn_asm_q256_bf16") != hipSuccess) d.fn_q256[F3R_BF16] = nullptr;
    }
    (void)hipGetLastError();
  }
  if (qk_planes == -256) return hd == 64 ? d.fn_q256[dtype] : nullptr;   // (internal code: the 256-query form of the head_dim-64 kernel)
  return qk_planes == 3 ? d.fn_qk3f8 : qk_planes == 2 ? d.fn_qk3 : d.fn[hi][dtype];
}

bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

int num_cus();

// Small launches at head_dim 64: 512-query work items do not always fill the chip -- N = 3 views are 96 items on 256 CUs.  The same kernel with two
// query blocks per wave (256-query items, f3r_attn_asm_q256_*) has twice the items; an item takes 0.545 of a 512-query item's time on random,
// sharply attending operands (profiles/r06_attn_q256_vs_q512_items_by_n.jsonl: N = 3 1.78x faster, N = 20 +7.5 %) but ~0.64 inside the model
// on near-uniform attention, where the 512-query kernel never leaves its fast path (fusion-only N = 20 with default-init weights: 65.6 ms against
// 63.0 ms, profiles/r06_q256_in_model_ab.jsonl).  So the form is taken only where the round count wins at the in-model ratio: launches of less
// than HALF a round of 512-query items (N <= 4 views: 24.8 -> 24.3 ms end to end at N = 3).
bool use_q256(const f3r_attn_args& a, int hd, int qkp) {
  if (hd != 64 || qkp != 1) return false;
  static const char* force = getenv("F3R_ATTN_Q256");   // measurement only: "0" = never, "1" = always (tools/kernel_bench.py); the product never sets it
  if (force && (force[0] == '0' || force[0] == '1')) return force[0] == '1' && get_fn(a.dtype, hd, -256) != nullptr;
  const int64_t cus = num_cus();
  const int64_t hb = (int64_t)a.n_heads * a.batch;
  const int64_t n512 = (a.tq + 511) / 512 * hb, n256 = (a.tq + 255) / 256 * hb;
  const double t512 = (double)((n512 + cus - 1) / cus), t256 = 0.64 * (double)((n256 + cus - 1) / cus);
  return t256 < 0.97 * t512 && get_fn(a.dtype, hd, -256) != nullptr;
}

int num_cus() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
  return cus;
}

}  // namespace

// does an (eligible) launch take the 256-query form of the head_dim-64 kernel?  (f3r_attn_kernel_name)
bool f3r_attn_asm_uses_q256(const f3r_attn_args& a) {
  const int hd = a.head_dim == 0 ? 64 : a.head_dim;
  return use_q256(a, hd, a.qk_planes >= 2 ? a.qk_planes : 1);
}

// Can the hand-scheduled kernel take this launch?  *why names the first obstacle.
bool f3r_attn_asm_eligible(const f3r_attn_args& a, int64_t min_keys, const char** why) {
  static const char* none = "";
  *why = none;
  const int hd = a.head_dim == 0 ? 64 : a.head_dim;
  if (hd_index(hd) < 0) { *why = "no generated kernel for this head_dim (64, 80, 128)"; return false; }
  const int qkp = a.qk_planes >= 2 ? a.qk_planes : 1;
  if (qkp >= 2 && (hd != 64 || a.dtype != F3R_F16)) { *why = "qk_planes 2 / 3 need head_dim 64 and fp16"; return false; }
  const int64_t wrows = wave_rows(hd, qkp);
  if (a.causal) { *why = "causal mask"; return false; }
  if (!a.q_prescaled) { *why = "q not pre-scaled"; return false; }
  if (a.tq < wrows) { *why = "fewer query rows than one wave takes (128 at head_dim 64, 64 otherwise)"; return false; }
  if (a.kv_group > 1 && !pow2(a.kv_group)) { *why = "kv_group not a power of two"; return false; }
  // (the three-product kernels index the parked state by sequence: rows [z tq, (z + 1) tq) -- they park it on every launch; the others at batch 1 only)
  if ((a.state_in || a.state_out) && a.batch != 1 && a.qk_planes < 2) { *why = "carried softmax state with batch > 1"; return false; }
  int first = -1;
  int64_t keys = 0;
  for (int s = 0; s < a.n_seg; ++s) {
    if (a.seg_len[s] == 0) continue;
    if (a.seg_len[s] % 64 != 0) { *why = "a K/V segment is not a multiple of 64 keys"; return false; }
    if (first < 0) first = s;
    if (a.ldvt[s] != a.ldvt[first] || a.k_batch_stride[s] != a.k_batch_stride[first] || a.vt_batch_stride[s] != a.vt_batch_stride[first]) {
      *why = "K/V segments with different ldvt / batch strides";
      return false;
    }
    keys += a.seg_len[s];
  }
  if (first < 0) { *why = "no keys"; return false; }
  if (keys < min_keys) { *why = "fewer keys than F3R_ATTN_ASM_MIN_KEYS"; return false; }
  if (keys / 64 >= (1ll << 31)) { *why = "too many keys"; return false; }
  // 32-bit lane offsets: a wave's query rows, 64 key rows, head_dim V^T rows must span < 4 GiB
  if (wrows * a.ldq * 2 >= (1ll << 32) || wrows * a.ldo * 2 >= (1ll << 32) || 64 * a.ldk * 2 >= (1ll << 32) || (int64_t)hd * a.ldvt[first] * 2 >= (1ll << 32) ||
      wrows * a.n_heads * hd * 4 >= (1ll << 32)) {
    *why = "row strides too large for 32-bit lane offsets";
    return false;
  }
  if (a.tq >= (1ll << 31) || a.n_heads >= 65536 || a.batch >= 65536) { *why = "grid too large"; return false; }
  // a code object that does not load on this device makes the launch INELIGIBLE (kernel_sel 0 then takes the general HIP kernel,
  // kernel_sel 2 reports why) instead of failing every fusion-attention call of the process
  if (get_fn(a.dtype, hd, qkp) == nullptr) { *why = "the embedded code object could not be loaded on this device"; return false; }
  return true;
}

// The last, partly filled round of a head_dim-64 launch as 256-query items (round 6).  A launch of n 512-query items on `cus` persistent workgroups
// takes ceil(n / cus) round times whatever the remainder r = n mod cus is: N = 20 views are 640 items = 2.5 rounds, N = 100 are 3 200 = 12.5 -- half
// the chip idles through the last round.  When r <= cus / 2 the query blocks of that remainder go to a SECOND launch, which the rule above (use_q256)
// then runs on the 256-query form: 2 r <= cus items of 0.64 round time each, i.e. the launch pair costs floor(n / cus) + 0.64 instead of
// floor(n / cus) + 1 (fusion-only N = 20: -12 % on paper, -5.8 % measured; N = 320 is 40 full rounds and is left alone).  The main
// launch keeps whole query blocks x all heads x all batches, so its item count is a multiple of cus exactly when its block count is a multiple of
// cus / gcd(cus, heads x batch).  Not with a carried softmax state (the sharded path compares its two-launch form against one launch), not
// with CUs reserved.  F3R_ATTN_TAIL_SPLIT=0 (measurement only) switches it off.
static int64_t tail_split_rows(const f3r_attn_args& a, int hd, int qkp) {
  if (hd != 64 || qkp != 1 || a.state_in || a.state_out || a.reserve_cus > 0) return 0;
  static const char* off = getenv("F3R_ATTN_TAIL_SPLIT");
  if (off && off[0] == '0') return 0;
  static const char* force = getenv("F3R_ATTN_Q256");
  if (force && (force[0] == '0' || force[0] == '1')) return 0;
  if (get_fn(a.dtype, hd, -256) == nullptr) return 0;
  const int64_t cus = num_cus(), hb = (int64_t)a.n_heads * a.batch;
  const int64_t nqb = (a.tq + 511) / 512;
  if (nqb * hb <= cus) return 0;   // less than a round: use_q256 decides for the whole launch
  int64_t x = cus, y = hb;
  while (y) { const int64_t t = x % y; x = y; y = t; }
  const int64_t g = cus / x;       // query blocks per whole number of rounds
  const int64_t nqb_main = nqb / g * g;
  if (nqb_main == 0 || nqb_main == nqb) return 0;
  // long launches: the end of the first launch is a chip-wide join, and the XCDs' clocks differ by up to 6 % under the power cap -- work stealing
  // inside ONE launch lets the fast ones take the last round's items instead.  Measured at N = 100 (12 whole rounds + 128 items): 33.44 ms against
  // 33.41 ms for one launch (profiles/r06_attn_last_round_split_ab.jsonl); fusion-only N = 20 (2 + 128): 1.540 against 1.635 ms.
  if (nqb_main * hb / cus > 8) return 0;
  const int64_t r = (nqb - nqb_main) * hb;   // items of the last round (< cus)
  if (2 * r > cus) return 0;                 // the tail would need two rounds of 256-query items: 1.28 > 1
  return nqb_main * 512;
}

bool f3r_attn_asm_splits_tail(const f3r_attn_args& a) {
  const int hd = a.head_dim == 0 ? 64 : a.head_dim;
  return tail_split_rows(a, hd, a.qk_planes >= 2 ? a.qk_planes : 1) > 0;
}

static int launch_one(const f3r_attn_args& a, hipStream_t stream);

int f3r_attn_asm_launch(const f3r_attn_args& a, hipStream_t stream) {
  const int hd = a.head_dim == 0 ? 64 : a.head_dim;
  const int64_t rows = tail_split_rows(a, hd, a.qk_planes >= 2 ? a.qk_planes : 1);
  if (rows <= 0) return launch_one(a, stream);
  f3r_attn_args m = a, t = a;
  m.tq = rows;
  t.tq = a.tq - rows;
  t.q = (const char*)a.q + rows * a.ldq * 2;
  t.o = (char*)a.o + rows * a.ldo * 2;
  const int rc = launch_one(m, stream);
  return rc != F3R_OK ? rc : launch_one(t, stream);
}

static int launch_one(const f3r_attn_args& a, hipStream_t stream) {
  const int hd = a.head_dim == 0 ? 64 : a.head_dim;
  const int qkp = a.qk_planes >= 2 ? a.qk_planes : 1;
  const bool q256 = use_q256(a, hd, qkp);
  hipFunction_t fn = get_fn(a.dtype, hd, q256 ? -256 : qkp);
  if (!fn) {
    f3r_set_error("f3r_attn_fwd: the embedded hand-scheduled kernel could not be loaded on this device");
    return F3R_ERR_LAUNCH;
  }
  f3r_attn_asm_args k;
  memset(&k, 0, sizeof(k));
  k.q = a.q;
  k.o = a.o;
  k.ldq_b = (uint32_t)(a.ldq * 2);
  k.ldk_b = (uint32_t)(a.ldk * 2);
  k.ldo_b = (uint32_t)(a.ldo * 2);
  k.q_bs = (uint64_t)a.q_batch_stride * 2;
  k.o_bs = (uint64_t)a.o_batch_stride * 2;
  int sh = 0;
  for (int gsz = a.kv_group > 1 ? a.kv_group : 1; gsz > 1; gsz >>= 1) ++sh;
  k.kv_shift = (uint32_t)sh;
  k.flags = (a.state_in ? 1u : 0u) | (a.state_out ? 2u : 0u);
  k.st_o = a.st_o;
  k.st_ml = a.st_ml;
  k.st_o_ld_b = (uint32_t)a.n_heads * (uint32_t)hd * 4u;
  k.st_ml_ld_b = (uint32_t)a.n_heads * 16u;
  k.tq = (uint32_t)a.tq;
  k.dbg = a.dbg_counters;
  for (int s = 0; s < a.n_seg; ++s) {
    if (a.seg_len[s] == 0) continue;
    f3r_attn_asm_seg& g = k.seg[k.n_seg++];
    g.k = a.k_seg[s];
    g.vt = a.vt_seg[s];
    g.tiles = (uint32_t)(a.seg_len[s] / 64);
    k.n_tiles += g.tiles;
    k.ldvt_b = (uint32_t)(a.ldvt[s] * 2);
    k.k_bs = (uint64_t)a.k_batch_stride[s] * 2;
    k.vt_bs = (uint64_t)a.vt_batch_stride[s] * 2;
  }
  // Work stealing (f3r_attn_args.sched_counter): one persistent workgroup per CU takes (q block, head, batch) items from a shared counter, so
  // the XCDs -- which the hardware feeds round-robin by workgroup id whatever their clocks -- finish together.  Worth it from two rounds of
  // workgroups on; the kernel's magic-number division needs item x period < 2^32.
  const int64_t wg_rows = q256 ? 256 : 4 * wave_rows(hd, qkp);
  const unsigned nx = (unsigned)((a.tq + wg_rows - 1) / wg_rows);
  unsigned gx = nx, gy = (unsigned)a.n_heads, gz = (unsigned)a.batch;
  const uint64_t n_work = (uint64_t)nx * gy * gz, nxy = (uint64_t)nx * gy;
  unsigned cus = (unsigned)num_cus();
  // f3r_attn_args.reserve_cus: leave that many CUs to whatever else must become resident while this launch runs (a rank's local-shard launch
  // next to the exchange of the other ranks' K / V^T); the persistent form is then taken whenever there are more items than workgroups
  const unsigned reserve = a.reserve_cus > 0 ? (unsigned)a.reserve_cus : 0u;
  const bool reserving = a.sched_counter && reserve > 0 && reserve < cus && n_work > (uint64_t)(cus - reserve);
  if (reserving) cus -= reserve;
  if (a.sched_counter && (reserving || n_work >= 2ull * cus) && n_work * nxy < (1ull << 32)) {
    k.sched = a.sched_counter;
    k.n_work = (uint32_t)n_work;
    k.nx = nx;
    k.nxy = (uint32_t)nxy;
    k.nx_magic = nx > 1 ? (uint32_t)(((1ull << 32) + nx - 1) / nx) : 0;
    k.nxy_magic = nxy > 1 ? (uint32_t)(((1ull << 32) + nxy - 1) / nxy) : 0;
    k.grid = cus;
    gx = cus;
    gy = gz = 1;
    // the kernel trusts {next, done} to be zero on entry and leaves them zero; a launch that was aborted (or a buffer the caller did not clear)
    // would make the next one silently skip items [0, next): clear the two words on the launch stream (a memset node under graph capture)
    hipError_t me = hipMemsetAsync(a.sched_counter, 0, 2 * sizeof(uint32_t), stream);
    if (me != hipSuccess) {
      f3r_set_error("f3r_attn_fwd: clearing sched_counter failed: %s", hipGetErrorString(me));
      return F3R_ERR_LAUNCH;
    }
  }
  size_t size = sizeof(k);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  hipError_t e = hipModuleLaunchKernel(fn, gx, gy, gz, 256, 1, 1, 0, stream, nullptr, config);
  if (e != hipSuccess) {
    f3r_set_error("f3r_attn_fwd: hipModuleLaunchKernel failed: %s", hipGetErrorString(e));
    return F3R_ERR_LAUNCH;
  }
  return f3r_check_launch("f3r_attn_fwd(asm)");
}
