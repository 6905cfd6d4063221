#!/bin/bash
# Round-2 closing run: the driver's three tiers (pytest -m gpu, smoke, default bench) + rocprofv3 stats of the bench command.
cd /root/repo; mkdir -p gpurun_out/r2f; export TMPDIR=/tmp
O=gpurun_out/r2f
( time timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 ) > $O/pytest_gpu.log 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_default.log 2>&1; grep '"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
timeout 300 python bench.py --views 100 --no-cpu-baseline > $O/bench_n100.log 2>&1; grep '"metric"' $O/bench_n100.log | tail -1 > $O/bench_n100.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_high -- python bench.py --no-alt --no-cpu-baseline --no-parity > $O/prof_high.log 2>&1
f=$(find $O/prof_high -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_high.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
tail -4 $O/pytest_gpu.log; tail -2 $O/smoke.log; cut -c1-300 $O/bench_default.json; echo; cut -c1-300 $O/bench_n100.json; echo; head -8 $O/kernel_stats_high.csv | cut -c1-200
