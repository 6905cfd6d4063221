#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE) of the attention kernel at a given fusion shape, separate --pmc passes.
VIEWS=${1:-320}; DT=${2:-fp16}
mkdir -p gpurun_out/traffic
export TMPDIR=/tmp
CMD="python tools/kernel_bench.py --what attnproduct --attn-dtypes $DT --views $VIEWS"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/traffic/f_$VIEWS --output-format csv -- $CMD > gpurun_out/traffic/f_$VIEWS.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/traffic/w_$VIEWS --output-format csv -- $CMD > gpurun_out/traffic/w_$VIEWS.log 2>&1
python - <<PY
import csv, glob, json
out = {}
for tag in ("f", "w"):
    f = glob.glob(f"gpurun_out/traffic/{tag}_$VIEWS/*/*counter_collection.csv")[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "attn_kernel" in r["Kernel_Name"]]
    out[tag] = sum(v) / len(v)
print(json.dumps({"views": $VIEWS, "operands": "$DT", "FETCH_SIZE_KiB": out["f"], "WRITE_SIZE_KiB": out["w"]}))
PY
find gpurun_out/traffic -name "*kernel_trace.csv" -delete
