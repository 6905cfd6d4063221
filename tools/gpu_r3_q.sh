#!/bin/bash
# round 3, GPU call Q (closing): the driver's three tiers on the final tree + the RCCL single-rank bench path + rocprofv3 stats of the default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3q; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -25 ) > $O/pytest_gpu.log 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
timeout 900 python bench.py > $O/bench_default.log 2>&1; grep '"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
F3R_BENCH_FORCE_DIST=1 timeout 300 python bench.py --views 20 --steps 1 --warmup 1 --no-cpu-baseline --no-alt > $O/bench_forced_dist.log 2>&1; echo "forced-dist exit $?" >> $O/bench_forced_dist.log
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_high -- python bench.py --no-alt --no-cpu-baseline --no-parity > $O/prof_high.log 2>&1
f=$(find $O/prof_high -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_high.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/prof_high
tail -6 $O/pytest_gpu.log; tail -2 $O/smoke.log; cut -c1-900 $O/bench_default.json; echo; tail -2 $O/bench_forced_dist.log | cut -c1-300; head -4 $O/kernel_stats_high.csv | cut -c1-200
