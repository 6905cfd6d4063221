#!/bin/bash
mkdir -p gpurun_out/b
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gemm256_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/b/pytest_gemm256.log )
tail -4 gpurun_out/b/pytest_gemm256.log
F3R_LAB_LIB=$PWD/tools/lab/libf3r_hip_lab.so timeout 600 python tools/kernel_bench.py --what lab > gpurun_out/b/lab.jsonl 2> gpurun_out/b/lab.err
cat gpurun_out/b/lab.jsonl | cut -c1-400; tail -3 gpurun_out/b/lab.err
timeout 600 python tools/kernel_bench.py --what gemm,conv > gpurun_out/b/kernel_bench.jsonl 2> gpurun_out/b/kernel_bench.err
