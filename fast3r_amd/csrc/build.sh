#!/bin/bash
# Builds fast3r_amd/lib/libf3r_hip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../lib"
mkdir -p "$out" "$here/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${F3R_EXTRA_FLAGS:-}"
pids=()
for f in f3r_gemm f3r_gemm256 f3r_gemm256_bf16 f3r_attn f3r_elem f3r_post f3r_pnp f3r_exact f3r_capi; do
  if [ ! -f "$here/obj/$f.o" ] || [ "$here/$f.hip" -nt "$here/obj/$f.o" ] || [ "$here/f3r_common.h" -nt "$here/obj/$f.o" ] || [ "$here/f3r_gemm_epi.h" -nt "$here/obj/$f.o" ] || [ "$here/f3r_gemm256_impl.h" -nt "$here/obj/$f.o" ] || [ "$here/f3r_linalg.h" -nt "$here/obj/$f.o" ] || [ "$0" -nt "$here/obj/$f.o" ] || [ "$here/../../include/f3r.h" -nt "$here/obj/$f.o" ]; then
    extra=""
    # attention: keep MFMA results in VGPRs (the softmax reads them with VALU instructions); with one wave per SIMD the
    # AGPR half of the register file then serves as spill space instead of scratch memory
    [ "$f" = f3r_attn ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
    $HIPCC $FLAGS $extra -c "$here/$f.hip" -o "$here/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC "$here"/obj/f3r_gemm.o "$here"/obj/f3r_gemm256.o "$here"/obj/f3r_gemm256_bf16.o "$here"/obj/f3r_attn.o "$here"/obj/f3r_elem.o "$here"/obj/f3r_post.o "$here"/obj/f3r_pnp.o "$here"/obj/f3r_exact.o "$here"/obj/f3r_capi.o -o "$out/libf3r_hip.so"
echo "built $out/libf3r_hip.so"
