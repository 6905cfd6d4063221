#!/bin/bash
# round 3, GPU call H: re-run of the fixed new tests + attention variants (fold via v_fma_mix, deeper LDS-DMA prefetch) + PMC evidence at N=320
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r3h; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_attn_asm_gpu.py -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -30 > $out/asm.log; tail -3 $out/asm.log
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -p no:cacheprovider -s -k "hd80 or anchored" 2>&1 | grep -v amdgpu.ids | tail -60 > $out/e2e_hd80.log; grep -E "parity\]|passed|failed" $out/e2e_hd80.log | tail -30
timeout 1200 python -m pytest tests/test_realsize_gpu.py -q -p no:cacheprovider -s 2>&1 | grep -v amdgpu.ids > $out/realsize.log; grep -E "parity\]|passed|failed|Error|assert" $out/realsize.log | tail -30
for v in mix pf3 mixpf3 pf4 mixg3; do
  echo "== $v" >> $out/attnsel_variants.jsonl
  F3R_LAB_LIB=tools/lab/var/libf3r_$v.so timeout 200 python tools/kernel_bench.py --what attnsel --views 100,320 --attn-dtypes fp16 --sels 2 2>&1 | grep -v amdgpu.ids >> $out/attnsel_variants.jsonl
done
echo "== g5 bf16" >> $out/attnsel_variants.jsonl
F3R_LAB_LIB=tools/lab/var/libf3r_g5.so timeout 200 python tools/kernel_bench.py --what attnsel --views 320 --attn-dtypes bf16 --sels 2 2>&1 | grep -v amdgpu.ids >> $out/attnsel_variants.jsonl
echo "== product" >> $out/attnsel_variants.jsonl
timeout 300 python tools/kernel_bench.py --what attnsel --views 100,320 --attn-dtypes fp16,bf16 --sels 2,1 2>&1 | grep -v amdgpu.ids >> $out/attnsel_variants.jsonl
cat $out/attnsel_variants.jsonl
