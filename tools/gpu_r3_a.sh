#!/bin/bash
# round 3, GPU call A: first light of the hand-scheduled attention kernel (parity, then timing next to the HIP kernel, then generator variants)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r3a; mkdir -p $out
timeout 120 python tools/attn_asm_check.py fp16 512 128 1 > $out/check.log 2>&1; rc=$?
echo "check rc $rc" >> $out/check.log; cat $out/check.log
if [ $rc -ne 0 ]; then
  timeout 300 rocgdb -batch -ex "set pagination off" -ex run -ex "info threads" -ex "x/8i \$pc-16" -ex "info registers pc s22 s8 s9 s10 s11 s12 s13 s14 s15 s28 m0 vcc" -ex "p \$v4" -ex "p \$v250" -ex "p \$v252" --args python tools/attn_asm_check.py fp16 512 128 1 > $out/gdb.log 2>&1
  tail -80 $out/gdb.log
  exit 0
fi
timeout 120 python tools/attn_asm_check.py fp16 1024 4096 2 >> $out/check.log 2>&1
timeout 120 python tools/attn_asm_check.py bf16 512 640 2 >> $out/check.log 2>&1
tail -8 $out/check.log
timeout 900 python -m pytest tests/test_attn_asm_gpu.py -x -q > $out/pytest_asm.log 2>&1; echo "pytest rc $?" >> $out/pytest_asm.log
tail -15 $out/pytest_asm.log
timeout 400 python tools/kernel_bench.py --what attnsel --views 20,100,320 --attn-dtypes fp16,bf16 > $out/attnsel.jsonl 2>&1
cat $out/attnsel.jsonl
for v in g4 g6 add5 add6; do
  echo "== $v" >> $out/attnsel_variants.jsonl
  F3R_LAB_LIB=tools/lab/var/libf3r_$v.so timeout 200 python tools/kernel_bench.py --what attnsel --views 100,320 --attn-dtypes fp16 --sels 2 >> $out/attnsel_variants.jsonl 2>&1
done
cat $out/attnsel_variants.jsonl
