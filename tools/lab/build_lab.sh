#!/bin/bash
# Builds tools/lab/libf3r_hip_lab.so: the product library with (a) the GEMM ablation variants compiled in (-DF3R_GEMM_LAB, kernel_sel >= 16)
# and (b) the attention variant study (tools/lab/f3r_attn_variants.hip, 90 variants, -DF3R_ATTN_LAB) in place of the product attention kernel.
# Used only by tools/kernel_bench.py through F3R_LAB_LIB; nothing in fast3r_amd/, tests/ or bench.py loads it.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
src="$here/../../fast3r_amd/csrc"
obj="${TMPDIR:-/tmp}/f3r_labobj"
mkdir -p "$obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DF3R_GEMM_LAB -I$src -I$here"
pids=()
for f in f3r_gemm f3r_gemm256 f3r_gemm256_bf16 f3r_elem f3r_post f3r_pnp f3r_exact f3r_capi; do
  $HIPCC $FLAGS -c "$src/$f.hip" -o "$obj/$f.o" & pids+=($!)
done
ATTN_FLAGS="-mllvm -amdgpu-mfma-vgpr-form"
[ "${F3R_LAB_ATTN_ALL:-0}" = 1 ] && ATTN_FLAGS="$ATTN_FLAGS -DF3R_ATTN_LAB"
$HIPCC $FLAGS $ATTN_FLAGS -c "$here/f3r_attn_variants.hip" -o "$obj/f3r_attn_variants.o" & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC "$obj"/f3r_gemm.o "$obj"/f3r_gemm256.o "$obj"/f3r_gemm256_bf16.o "$obj"/f3r_elem.o "$obj"/f3r_post.o "$obj"/f3r_pnp.o "$obj"/f3r_exact.o "$obj"/f3r_capi.o "$obj"/f3r_attn_variants.o -o "$here/libf3r_hip_lab.so"
echo "built $here/libf3r_hip_lab.so"
