#!/bin/bash
# round 3, GPU call X: fillers per MFMA gap in the hand-scheduled attention kernel (product: 4 for fp16 / 5 for bf16, which leaves 20-30 softmax
# instructions after the last MFMA of a stage) against 5 / 6 and mixed patterns, T = 327 680, same box, same call
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3x; mkdir -p $O; export TMPDIR=/tmp
run() { name=$1; dt=$2
  if [ $name = product ]; then unset F3R_LAB_LIB; else export F3R_LAB_LIB=$PWD/tools/lab/var/libf3r_$name.so; fi
  echo "== $name $dt" >> $O/sweep.jsonl
  timeout 200 python tools/kernel_bench.py --what attnsel --attn-dtypes $dt --views 320 --sels 2 >> $O/sweep.jsonl 2>> $O/err.log
}
run product fp16; run bg5 fp16; run bg45 fp16; run bg455 fp16; run product fp16
run product bf16; run bg6 bf16; run bg56 bf16
cat $O/sweep.jsonl | cut -c1-200
