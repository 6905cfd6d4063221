#!/bin/bash
# rocprofv3 kernel-trace + stats of one bench configuration; summaries land in gpurun_out/prof_<tag>/
TAG=${1:-n100}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python bench.py "$@" > gpurun_out/prof_$TAG.log 2>&1
echo "exit $?" >> gpurun_out/prof_$TAG.log
grep '"metric"' gpurun_out/prof_$TAG.log | tail -1
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); echo "stats: $f"; head -30 "$f"
# drop the bulky per-dispatch trace, keep the stats
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +8M -delete
