#!/bin/bash
# Builds tools/lab/libf3r_hip_lab.so: the product library with the ablation variants of the 8-wave GEMM compiled in (-DF3R_GEMM_LAB,
# kernel_sel >= 16).  Used only by tools/kernel_bench.py --what lab / labtime through F3R_LAB_LIB; nothing in fast3r_amd/, tests/ or bench.py
# loads it.  (The 90-variant study of the HIP attention kernel of rounds 1 - 2 -- f3r_attn_lab.h, f3r_attn_variants.hip, f3r_attn_xp.h -- was
# removed in round 4: its conclusions are in docs/history/ (the lab notes of rounds 1 - 4), its sources in git history up to commit f40f83d.  Variants of the
# hand-scheduled kernels: tools/lab/build_attn_variants.sh, tools/gemm_lab.py.)
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
src="$here/../../fast3r_amd/csrc"
obj="${TMPDIR:-/tmp}/f3r_labobj"
mkdir -p "$obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DF3R_GEMM_LAB -I$src -I$here"
pids=()
[ -f "$src/obj/f3r_attn_asm_blob.o" ] || "$src/build.sh"   # the embedded code objects come from the product build
for f in f3r_gemm f3r_gemm256 f3r_gemm256_bf16 f3r_gemm256_f8 f3r_robust f3r_gemm_asm f3r_attn_asm f3r_attn_generic f3r_elem f3r_post f3r_pnp f3r_exact f3r_exact_mfma f3r_capi; do
  $HIPCC $FLAGS -c "$src/$f.hip" -o "$obj/$f.o" & pids+=($!)
done
$HIPCC $FLAGS -mllvm -amdgpu-mfma-vgpr-form -c "$src/f3r_attn.hip" -o "$obj/f3r_attn.o" & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC "$obj"/*.o "$src/obj/f3r_attn_asm_blob.o" "$src/obj/f3r_gemm_asm_blob.o" -o "$here/libf3r_hip_lab.so"
echo "built $here/libf3r_hip_lab.so"
