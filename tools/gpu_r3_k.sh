#!/bin/bash
# round 3, GPU call K: sharded-path tests (both exchange forms), emulated per-rank steps (N=320 and N=1500 of 8 ranks), quick N=320 / N=100 steps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r3k; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -p no:cacheprovider -k "sharded or rccl" 2>&1 | grep -v amdgpu.ids | tail -15 > $out/sharded_tests.log; tail -4 $out/sharded_tests.log
timeout 600 python bench.py --emulate-rank 3 --of 8 --steps 2 --warmup 1 --no-alt > $out/emu_n320_allgather.json 2> $out/emu1.err; cat $out/emu_n320_allgather.json
timeout 600 python bench.py --emulate-rank 3 --of 8 --steps 2 --warmup 1 --no-alt --exchange p2p > $out/emu_n320_p2p.json 2> $out/emu2.err; cat $out/emu_n320_p2p.json
timeout 900 python bench.py --emulate-rank 3 --of 8 --views 1500 --steps 1 --warmup 0 --no-alt > $out/emu_n1500_allgather.json 2> $out/emu3.err; cat $out/emu_n1500_allgather.json
timeout 600 python bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline --no-parity > $out/bench_n320_quick.json 2> $out/b1.err; cat $out/bench_n320_quick.json
timeout 600 python bench.py --views 100 --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $out/bench_n100_quick.json 2> $out/b2.err; cat $out/bench_n100_quick.json
