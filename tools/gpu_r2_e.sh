#!/bin/bash
mkdir -p gpurun_out/e
export TMPDIR=/tmp
O=gpurun_out/e
( timeout 600 python -m pytest tests/test_gemm256_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -15 > $O/pytest_gemm256.log ); tail -4 $O/pytest_gemm256.log
timeout 600 python tools/kernel_bench.py --what conv > $O/kernel_bench_conv.jsonl 2> $O/kb.err; cat $O/kernel_bench_conv.jsonl | cut -c1-330
