#!/bin/bash
# round 3, GPU call V: is the fp16 kernel's extra fabric traffic tied to its packed-fp16 row sums?  FETCH_SIZE + time of the product (pkadd) and of an
# fp32-add row-sum build, fp16 operands, T = 327 680
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3v; mkdir -p $O; export TMPDIR=/tmp
for name in product v2add; do
  if [ $name = product ]; then unset F3R_LAB_LIB; else export F3R_LAB_LIB=$PWD/tools/lab/var/libf3r_$name.so; fi
  timeout 200 python tools/kernel_bench.py --what attnsel --attn-dtypes fp16 --views 320 --sels 2 > $O/t_$name.jsonl 2>> $O/err.log
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f_$name --output-format csv -- python tools/kernel_bench.py --what attnsel --attn-dtypes fp16 --views 320 --sels 2 > $O/f_$name.log 2>&1
done
python - <<PY
import csv, glob, json
for name in ("product", "v2add"):
    f = glob.glob("$O/f_%s/*/*counter_collection.csv" % name)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "attn" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    print(name, "GB per dispatch (x2 corrected):", [round(2 * 1024 * v / 1e9, 1) for v in vals], open("$O/t_%s.jsonl" % name).read().strip()[-120:])
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
