#!/bin/bash
# round 3, GPU call O: emulated per-rank step of the N=320 forward for 2 and 4 ranks (8 ranks: call K) -> the compute side of the strong-scaling curve
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3o; mkdir -p $O; export TMPDIR=/tmp
for W in 2 4; do
  timeout 600 python bench.py --emulate-rank 1 --of $W --steps 2 --warmup 1 --no-alt > $O/emu_n320_rank1of$W.json 2> $O/emu$W.err; cut -c1-700 $O/emu_n320_rank1of$W.json; echo
done
timeout 300 python tools/small_n_latency.py --views 2,3,8,20,40 > $O/small_n.jsonl 2> $O/small.err; tail -12 $O/small_n.jsonl
