#!/bin/bash
# round 3, GPU call B: parity of the hand-scheduled kernel (row sums by f32 adds), MFMA-gap micro-benchmark, ablations
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r3b; mkdir -p $out
for c in "fp16 512 128 1" "fp16 1024 4096 2" "bf16 512 640 2"; do timeout 120 python tools/attn_asm_check.py $c >> $out/check.log 2>&1; done
grep -v amdgpu.ids $out/check.log | tail -12
timeout 900 python -m pytest tests/test_attn_asm_gpu.py -q > $out/pytest_asm.log 2>&1; echo "pytest rc $?" >> $out/pytest_asm.log
tail -25 $out/pytest_asm.log
timeout 300 python tools/ubench/gap_ubench.py /tmp/ub > $out/gap_ubench.jsonl 2>&1
cat $out/gap_ubench.jsonl
timeout 400 python tools/kernel_bench.py --what attnsel --views 100,320 --attn-dtypes fp16,bf16 > $out/attnsel.jsonl 2>&1
grep -v amdgpu.ids $out/attnsel.jsonl
for v in nosm nosm_nodma nosm_nobar nosm_nodma_nobar nok8 nodma nobar nosum nocvt noexp dot5 add4 add7; do
  echo "== $v" >> $out/attnsel_variants.jsonl
  F3R_LAB_LIB=tools/lab/var/libf3r_$v.so timeout 200 python tools/kernel_bench.py --what attnsel --views 100 --attn-dtypes fp16 --sels 2 2>&1 | grep -v amdgpu.ids >> $out/attnsel_variants.jsonl
done
cat $out/attnsel_variants.jsonl
