#!/bin/bash
# round 3, GPU call J: persistent 256-tile GEMM: all GEMM / conv kernel tests, then persistent vs one-tile-per-workgroup timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r3j; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm256_gpu.py tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -q -p no:cacheprovider -x 2>&1 | grep -v amdgpu.ids | tail -30 > $out/gemm_tests.log; tail -5 $out/gemm_tests.log
timeout 900 python tools/kernel_bench.py --what gemmpersist 2>&1 | grep -v amdgpu.ids > $out/gemm_persist.jsonl; cat $out/gemm_persist.jsonl
