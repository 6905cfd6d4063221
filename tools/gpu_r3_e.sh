#!/bin/bash
# round 3, GPU call E: whole GPU suite (incl. the real-size parity tests) + the default bench line with the hand-scheduled attention kernel in place
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r3e; mkdir -p $out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q -rA -p no:cacheprovider -x 2>&1 | grep -v amdgpu.ids | tail -120 > $out/pytest_gpu.log
echo "pytest exit: $?" >> $out/pytest_gpu.log
grep -E "passed|failed|error|parity\]" $out/pytest_gpu.log | tail -40
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "bench exit: $?"; cat $out/bench_default.json
