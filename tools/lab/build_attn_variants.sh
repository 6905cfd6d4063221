#!/bin/bash
# Measurement builds of the hand-scheduled attention kernel: one library per generator setting, loaded by tools/kernel_bench.py through
# F3R_LAB_LIB.  Usage: build_attn_variants.sh NAME "GENERATOR FLAGS" [NAME "FLAGS" ...]  -> tools/lab/var/libf3r_NAME.so
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
src="$here/../../fast3r_amd/csrc"
LLVM="${LLVM_BIN:-/opt/rocm/lib/llvm/bin}"
mkdir -p "$here/var"
[ -f "$src/obj/f3r_capi.o" ] || "$src/build.sh"
while [ $# -ge 2 ]; do
  name="$1"; flags="$2"; shift 2
  w="${TMPDIR:-/tmp}/f3r_var_$name"; mkdir -p "$w"
  python3 "$src/asm/attn_gen.py" "$w/a.s" $flags   # (one generator since round 4; --layout is accepted and ignored)
  "$LLVM/clang" -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$w/a.s" -o "$w/a.o"
  "$LLVM/ld.lld" -shared "$w/a.o" -o "$w/a.hsaco"
  python3 - "$w/a.hsaco" "$w/blob.cpp" <<'PY'
import sys
b = open(sys.argv[1], "rb").read()
with open(sys.argv[2], "w") as f:
    f.write('extern "C" { extern const unsigned char f3r_attn_asm_hsaco[]; extern const unsigned int f3r_attn_asm_hsaco_len; }\n')
    f.write('alignas(4096) const unsigned char f3r_attn_asm_hsaco[] = {\n')
    for i in range(0, len(b), 32):
        f.write(",".join(str(x) for x in b[i:i + 32]) + ",\n")
    f.write('};\nconst unsigned int f3r_attn_asm_hsaco_len = %d;\n' % len(b))
PY
  g++ -O1 -fPIC -std=c++17 -c "$w/blob.cpp" -o "$w/blob.o"
  objs=""
  for f in f3r_gemm f3r_gemm256 f3r_gemm256_bf16 f3r_gemm_asm f3r_gemm_asm_blob f3r_attn f3r_attn_asm f3r_attn_generic f3r_elem f3r_post f3r_pnp f3r_exact f3r_exact_mfma f3r_robust f3r_capi; do objs="$objs $src/obj/$f.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs "$w/blob.o" -o "$here/var/libf3r_$name.so"
  echo "built $here/var/libf3r_$name.so ($flags)"
done
