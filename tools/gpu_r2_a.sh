#!/bin/bash
# Round-2 GPU call A: new GEMM kernel parity + regression of the kernel tests + kernel micro-benchmarks + golden e2e in all modes + a quick N=100 bench.
mkdir -p gpurun_out/a
export TMPDIR=/tmp
O=gpurun_out/a
( timeout 600 python -m pytest tests/test_gemm256_gpu.py -q -rA -p no:cacheprovider 2>&1 | tail -120 > $O/pytest_gemm256.log; echo "exit $?" >> $O/pytest_gemm256.log ) 
tail -3 $O/pytest_gemm256.log
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider 2>&1 | tail -40 > $O/pytest_kernels.log )
tail -2 $O/pytest_kernels.log
timeout 600 python tools/kernel_bench.py --what gemm,conv > $O/kernel_bench.jsonl 2> $O/kernel_bench.err
tail -3 $O/kernel_bench.err
( timeout 900 python -m pytest tests/test_e2e_gpu.py -q -rA -p no:cacheprovider -k "golden or api or rng or chunk or graph" 2>&1 | tail -150 > $O/pytest_e2e.log )
grep -c PASSED $O/pytest_e2e.log; grep "parity\]" $O/pytest_e2e.log | head -40; tail -3 $O/pytest_e2e.log
timeout 600 python bench.py --views 100 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_n100_bf16_fast.json 2> $O/bench_n100.err
tail -c 600 $O/bench_n100_bf16_fast.json
