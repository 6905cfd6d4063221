#!/bin/bash
# round 3, GPU call R: N=100 default bench + kernel stats; N=1500 on ONE GPU (cold, one step)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3r; mkdir -p $O; export TMPDIR=/tmp
timeout 400 python bench.py --views 100 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_n100.log 2>&1; grep '"metric"' $O/bench_n100.log | tail -1 > $O/bench_n100.json; cut -c1-400 $O/bench_n100.json; echo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_n100 -- python bench.py --views 100 --steps 3 --warmup 1 --no-alt --no-cpu-baseline --no-parity > $O/prof_n100.log 2>&1
f=$(find $O/prof_n100 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_n100.csv; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/prof_n100
head -6 $O/kernel_stats_n100.csv | cut -c1-160
timeout 600 python bench.py --views 1500 --steps 1 --warmup 0 --no-alt --no-cpu-baseline --no-parity > $O/bench_n1500.log 2>&1; grep '"metric"' $O/bench_n1500.log | tail -1 > $O/bench_n1500.json; cut -c1-500 $O/bench_n1500.json; echo
