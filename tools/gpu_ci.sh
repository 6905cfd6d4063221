#!/bin/bash
# One gpurun call: GPU test-suite (all failures reported), smoke, small bench.  Logs land in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocm-smi" > gpurun_out/env.log; rocm-smi --showproductname 2>&1 | head -20 >> gpurun_out/env.log; nproc >> gpurun_out/env.log
timeout 1500 python -m pytest tests -m gpu -q -rA -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit: $?" >> gpurun_out/smoke.log
timeout 600 python bench.py --views ${BENCH_VIEWS:-20} --steps 2 --warmup 1 > gpurun_out/bench_small.log 2>&1
echo "bench exit: $?" >> gpurun_out/bench_small.log
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -3 gpurun_out/bench_small.log
# the distributed code path of bench.py (RCCL init, view sharding, barrier, MAX all-reduce) in a world of one rank
F3R_BENCH_FORCE_DIST=1 timeout 600 python bench.py --views ${BENCH_VIEWS:-20} --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_forced_dist.log 2>&1
echo "forced-dist bench exit: $?" >> gpurun_out/bench_forced_dist.log
tail -2 gpurun_out/bench_forced_dist.log
