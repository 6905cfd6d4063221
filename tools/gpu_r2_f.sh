#!/bin/bash
mkdir -p gpurun_out/f
export TMPDIR=/tmp
O=gpurun_out/f
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm256_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -30 > $O/pytest_kernels.log ); tail -6 $O/pytest_kernels.log
( timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q -rA -p no:cacheprovider -k "golden or sharded or rccl or graph" 2>&1 | tail -120 > $O/pytest_e2e.log ); grep -E "gqa|portrait" $O/pytest_e2e.log | grep parity; tail -4 $O/pytest_e2e.log; grep FAILED $O/pytest_e2e.log | head
