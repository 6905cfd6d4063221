#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r3d; mkdir -p $out
for v in nok8 nok8pk nok8pk4 norare pk3 pk4 pk4k1 pk4k4; do
  echo "== $v" >> $out/attnsel_variants.jsonl
  F3R_LAB_LIB=tools/lab/var/libf3r_$v.so timeout 200 python tools/kernel_bench.py --what attnsel --views 100,320 --attn-dtypes fp16 --sels 2 2>&1 | grep -v amdgpu.ids >> $out/attnsel_variants.jsonl
done
timeout 200 python tools/kernel_bench.py --what attnsel --views 100,320 --attn-dtypes fp16 --sels 2,1 2>&1 | grep -v amdgpu.ids >> $out/attnsel_variants.jsonl
cat $out/attnsel_variants.jsonl
