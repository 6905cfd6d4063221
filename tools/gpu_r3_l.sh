#!/bin/bash
# round 3, GPU call L: the driver's three tiers (pytest -m gpu, smoke, default bench), PMC evidence for the fusion-attention kernel
# (both formats, T = 327 680), and the rocprofv3 --kernel-trace --stats summary of the default bench command.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3l; mkdir -p $O; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -25 ) > $O/pytest_gpu.log 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
timeout 900 python bench.py > $O/bench_default.log 2>&1; grep '"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
timeout 900 bash tools/pmc_r03_attn.sh 2 320 > $O/pmc.log 2>&1
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_high -- python bench.py --no-alt --no-cpu-baseline --no-parity > $O/prof_high.log 2>&1
f=$(find $O/prof_high -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_high.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/prof_high
tail -6 $O/pytest_gpu.log; tail -2 $O/smoke.log; cut -c1-1500 $O/bench_default.json; echo; tail -3 $O/pmc.log | cut -c1-1500; head -8 $O/kernel_stats_high.csv | cut -c1-200
