#!/bin/bash
# Round-2 evidence run: default bench (fp16/high + the bf16/fast alternative), rocprofv3 stats of the same command, PMC passes.
cd /root/repo; mkdir -p gpurun_out/r2h; export TMPDIR=/tmp
O=gpurun_out/r2h
timeout 600 python bench.py > $O/bench_default.log 2>&1; grep '"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json
timeout 300 python bench.py --views 100 --no-cpu-baseline > $O/bench_n100.log 2>&1; grep '"metric"' $O/bench_n100.log | tail -1 > $O/bench_n100.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_high -- python bench.py --no-alt --no-cpu-baseline --no-parity > $O/prof_high.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fast -- python bench.py --dtype bf16 --precision fast --no-alt --no-cpu-baseline --no-parity > $O/prof_fast.log 2>&1
for t in high fast; do f=$(find $O/prof_$t -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$t.csv; grep '"metric"' $O/prof_$t.log | tail -1 > $O/prof_${t}_bench.json; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
timeout 300 bash tools/pmc_mfma_util.sh fp16 320 > $O/pmc_util_fp16.log 2>&1
timeout 300 bash tools/pmc_mfma_util.sh bf16 320 > $O/pmc_util_bf16.log 2>&1
timeout 300 bash tools/pmc_traffic.sh 320 fp16 > $O/pmc_traffic_fp16.log 2>&1
timeout 400 bash tools/pmc_gemm.sh > $O/pmc_gemm.log 2>&1
cp gpurun_out/pmcu/*.json gpurun_out/pmcg/gemm_pmc.json $O/ 2>/dev/null
tail -3 $O/bench_default.json $O/bench_n100.json; tail -2 $O/pmc_util_fp16.log $O/pmc_util_bf16.log $O/pmc_traffic_fp16.log; tail -12 $O/pmc_gemm.log
head -12 $O/kernel_stats_high.csv
