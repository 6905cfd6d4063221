#!/bin/bash
# round 3, GPU call U: FETCH_SIZE of the fusion-attention kernel over repeated runs (how much the fabric-side traffic moves between runs of the same code)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3u; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2 3; do for V in fp16 bf16; do
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f_${V}_$i --output-format csv -- python tools/kernel_bench.py --what attnsel --attn-dtypes $V --views 320 --sels 2 > $O/f_${V}_$i.log 2>&1
done; done
python - <<PY
import csv, glob, json
out = {}
for f in sorted(glob.glob("$O/f_*_*/*/*counter_collection.csv")):
    key = f.split("/")[-3]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "attn" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    out[key] = {"dispatches": len(vals), "FETCH_SIZE_KiB_per_dispatch": [round(v) for v in vals], "bytes_x2_correction_GB_mean": round(2 * 1024 * sum(vals) / len(vals) / 1e9, 1)}
json.dump(out, open("$O/fetch_repeats.json", "w"), indent=1)
print(json.dumps({k: (v["bytes_x2_correction_GB_mean"], [round(2*1024*x/1e9,1) for x in v["FETCH_SIZE_KiB_per_dispatch"]]) for k, v in out.items()}))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
