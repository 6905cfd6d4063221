#!/bin/bash
# Builds tools/lab/libf3r_hip_convabl.so: the product library whose fused-tail x3f8 convolution launch can run with parts of its K loop removed
# (-DF3R_CONV_ABLATIONS; F3R_CONV_ABLATE=<bits> at run time, see f3r_gemm256_f8.hip) -- TIMING ONLY.  Used through F3R_LAB_LIB by
# tools/conv_f8_ab.py --roles head2 --splits x3f8; nothing in fast3r_amd/, tests/ or bench.py loads it.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
src="$here/../../fast3r_amd/csrc"
[ -f "$src/obj/f3r_gemm256.o" ] || "$src/build.sh"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DF3R_CONV_ABLATIONS -c "$src/f3r_gemm256_f8.hip" -o "${TMPDIR:-/tmp}/f3r_gemm256_f8_abl.o"
objs=()
for f in f3r_gemm f3r_gemm256 f3r_gemm256_bf16 f3r_gemm_asm f3r_gemm_asm_blob f3r_attn f3r_attn_asm f3r_attn_asm_blob f3r_attn_generic f3r_elem f3r_post f3r_pnp f3r_exact f3r_exact_mfma f3r_robust f3r_capi; do objs+=("$src/obj/$f.o"); done
$HIPCC --offload-arch=gfx950 -shared -fPIC "${objs[@]}" "${TMPDIR:-/tmp}/f3r_gemm256_f8_abl.o" -o "$here/libf3r_hip_convabl.so"
echo "built $here/libf3r_hip_convabl.so"
