#!/bin/bash
# round 3, GPU call I: second layout of the hand-scheduled attention kernel (no bias-step MFMAs): parity through the lab library, timing, PMC
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r3i; mkdir -p $out
export TMPDIR=/tmp
F3R_LAB_LIB=tools/lab/var/libf3r_v2.so timeout 600 python -m pytest tests/test_attn_asm_gpu.py -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -30 > $out/asm_v2.log; tail -4 $out/asm_v2.log
F3R_LAB_LIB=tools/lab/var/libf3r_v2pf3.so timeout 600 python -m pytest tests/test_attn_asm_gpu.py -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -30 > $out/asm_v2pf3.log; tail -2 $out/asm_v2pf3.log
for v in v2 v2pf3 v2mix; do
  echo "== $v" >> $out/attnsel_variants.jsonl
  F3R_LAB_LIB=tools/lab/var/libf3r_$v.so timeout 200 python tools/kernel_bench.py --what attnsel --views 100,320 --attn-dtypes fp16,bf16 --sels 2 2>&1 | grep -v amdgpu.ids >> $out/attnsel_variants.jsonl
done
echo "== product" >> $out/attnsel_variants.jsonl
timeout 300 python tools/kernel_bench.py --what attnsel --views 100,320 --attn-dtypes fp16,bf16 --sels 2,1 2>&1 | grep -v amdgpu.ids >> $out/attnsel_variants.jsonl
cat $out/attnsel_variants.jsonl
F3R_LAB_LIB=tools/lab/var/libf3r_v2.so timeout 600 tools/pmc_r03_attn.sh 2 320 2>&1 | tail -3
mkdir -p $out/pmc_v2; cp gpurun_out/pmc_r03/*.json $out/pmc_v2/
timeout 600 tools/pmc_r03_attn.sh 2 320 2>&1 | tail -3
mkdir -p $out/pmc_v1; cp gpurun_out/pmc_r03/*.json $out/pmc_v1/
