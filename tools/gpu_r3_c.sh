#!/bin/bash
# round 3, GPU call C: MFMA-gap micro-benchmark, row-sum / DMA variants of the hand-scheduled kernel, PMC (utilisation in cycles + clock)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r3c; mkdir -p $out /tmp/ub
timeout 300 python tools/ubench/gap_ubench.py /tmp/ub 2>&1 | grep -v amdgpu.ids > $out/gap_ubench.jsonl
cat $out/gap_ubench.jsonl
for v in pk5 pk4 pkrtz nt dmaearly dmalate nolds; do
  echo "== $v" >> $out/attnsel_variants.jsonl
  F3R_LAB_LIB=tools/lab/var/libf3r_$v.so timeout 200 python tools/kernel_bench.py --what attnsel --views 100,320 --attn-dtypes fp16 --sels 2 2>&1 | grep -v amdgpu.ids >> $out/attnsel_variants.jsonl
done
timeout 200 python tools/kernel_bench.py --what attnsel --views 100,320 --attn-dtypes fp16 --sels 2,1 2>&1 | grep -v amdgpu.ids >> $out/attnsel_variants.jsonl
cat $out/attnsel_variants.jsonl
timeout 300 tools/pmc_attn_asm.sh product fp16 100 2
F3R_LAB_LIB=tools/lab/var/libf3r_nosm.so timeout 300 tools/pmc_attn_asm.sh nosm fp16 100 2
F3R_LAB_LIB=tools/lab/var/libf3r_pk5.so timeout 300 tools/pmc_attn_asm.sh pk5 fp16 100 2
timeout 300 tools/pmc_attn_asm.sh hip fp16 100 1
cp gpurun_out/pmca/attn_*.json $out/
