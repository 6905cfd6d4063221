// TEST INFRASTRUCTURE: the product's SQPnP routine (fast3r_amd/csrc/f3r_sqpnp.h) compiled for the host, so that tests/test_pnp.py can
// check its arithmetic against the fp64 restatement (oracle/sqpnp.py) without a GPU.  Nothing in the product links this file.
#define F3R_HOST_BUILD 1
#include "../../fast3r_amd/csrc/f3r_sqpnp.h"

extern "C" int sqpnp_host_solve(const double* M, const double* xy, int n, double unit2, double* R9, double* t3, double* err) {
  double s[f3r_sqpnp::N_SUMS];
  for (int k = 0; k < f3r_sqpnp::N_SUMS; ++k) s[k] = 0.0;
  for (int i = 0; i < n; ++i) f3r_sqpnp::accumulate(s, M + 3 * i, xy[2 * i], xy[2 * i + 1]);
  f3r_sqpnp::Result r;
  if (!f3r_sqpnp::solve(s, unit2, r)) return 0;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R9[3 * i + j] = r.R[i][j];
    t3[i] = r.t[i];
  }
  *err = r.err;
  return 1;
}
