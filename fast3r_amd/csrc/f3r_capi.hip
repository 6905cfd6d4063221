// Host-side glue of libf3r_hip.so: version, per-thread last-error string, launch check.
#include <stdarg.h>
#include <stdio.h>

#include "f3r_common.h"

namespace {
thread_local char g_err[512] = "";
}

void f3r_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int f3r_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    f3r_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return F3R_ERR_LAUNCH;
  }
  return F3R_OK;
}

extern "C" int f3r_version(void) { return 350; /* 0.3.5: round-6 ABI (include/f3r.h f3r_version) */ }

extern "C" int f3r_wall_clock_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return khz;
}

extern "C" const char* f3r_last_error_string(void) { return g_err; }

extern "C" size_t f3r_sizeof(int what) {
  switch (what) {
    case 0: return sizeof(f3r_gemm_args);
    case 1: return sizeof(f3r_attn_args);
    case 2: return sizeof(f3r_attn_f32_args);
    default: return 0;
  }
}

extern "C" size_t f3r_block_workspace_bytes_ex(int64_t tokens, int D, int kv_dim, int hidden, int64_t n_seq, int64_t seq_len, int f8_rows, size_t offsets[5]) {
  if (tokens < 0 || D <= 0 || kv_dim < 0 || hidden <= 0 || n_seq <= 0 || seq_len < 0 || n_seq * seq_len != tokens || !offsets) return 0;
  if (f8_rows && (D % 8 != 0 || hidden % 8 != 0)) return 0;
  const size_t al = 256;
  const size_t ldvt = (size_t)((seq_len + 63) / 64 * 64);
  // f8_rows: regions 0 and 4 hold rows [w fp16 | w fp8] (3 w bytes) -- the LayerNorm output and the MLP hidden state of the F3R_SPLIT_W2F8 GEMMs; the
  // plain [tokens][w] forms (attention output, hidden state of a pass that stays on fp16 planes) alias the head of the same regions
  const size_t wide = f8_rows ? 3 : 2;
  const size_t sizes[5] = {(size_t)tokens * D * wide, (size_t)tokens * D * 2, (size_t)tokens * kv_dim * 2, (size_t)n_seq * kv_dim * ldvt * 2,
                           (size_t)tokens * hidden * wide};
  size_t off = 0;
  for (int i = 0; i < 5; ++i) {
    offsets[i] = off;
    off += (sizes[i] + al - 1) / al * al;
  }
  return off > 0 ? off : al;
}

extern "C" size_t f3r_block_workspace_bytes(int64_t tokens, int D, int kv_dim, int hidden, int64_t n_seq, int64_t seq_len, size_t offsets[5]) {
  return f3r_block_workspace_bytes_ex(tokens, D, kv_dim, hidden, n_seq, seq_len, 0, offsets);
}
