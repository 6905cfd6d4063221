#!/bin/bash
# Round-2 GPU call C: whole GPU test-suite + smoke + headline bench in both modes + rocprofv3 kernel stats
mkdir -p gpurun_out/c
export TMPDIR=/tmp
O=gpurun_out/c
( timeout 1500 python -m pytest tests -m gpu -q -rA -p no:cacheprovider 2>&1 | tail -260 > $O/pytest_gpu.log; echo "pytest exit: $?" >> $O/pytest_gpu.log )
grep "parity\]" $O/pytest_gpu.log | grep -v "tiny_" ; grep -E "passed|failed" $O/pytest_gpu.log | tail -3; grep FAILED $O/pytest_gpu.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --dtype bf16 --precision fast --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_n320_bf16_fast.json 2> $O/bench_n320_bf16_fast.err
timeout 900 python bench.py --dtype fp16 --precision high --steps 2 --warmup 1 > $O/bench_n320_fp16_high.json 2> $O/bench_n320_fp16_high.err
timeout 900 python bench.py --dtype fp16 --precision fast --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_n320_fp16_fast.json 2> $O/bench_n320_fp16_fast.err
for f in $O/bench_n320_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],2), round(d['ms_per_step']), round(d['roofline']['achieved']), round(d['roofline']['e2e']['frac'],3), d.get('parity',{}).get('rel_l2'), d.get('cpu_baseline'))"; done
R=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_fp16_high -o r02 -- python $R/bench.py --dtype fp16 --precision high --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $R/$O/prof_bench.json 2> $R/$O/prof_bench.err; cd $R
find $O/prof_fp16_high -name "*kernel_stats*" | head; ls -la $O/prof_fp16_high/* | head
find $O/prof_fp16_high -name "*kernel_trace*" -size +1M -delete
