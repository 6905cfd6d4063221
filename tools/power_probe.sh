#!/bin/bash
# Samples GPU clock / power while an attention micro-benchmark runs (is the kernel power-limited?).
# usage (on the GPU box): tools/power_probe.sh <variant> [views]
export TMPDIR=/tmp
v=${1:-55}; n=${2:-320}
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.4; done ) > gpurun_out/power_v${v}.txt &
sampler=$!
python tools/kernel_bench.py --what attnonly --views $n --variants $v 2>&1 | grep tflops
kill $sampler 2>/dev/null
wait $sampler 2>/dev/null
sort gpurun_out/power_v${v}.txt | uniq -c | sort -rn | head -8
