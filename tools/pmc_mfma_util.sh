#!/bin/bash
# rocprofv3 PMC: matrix-pipe utilisation of the product attention kernel in CYCLES (independent of the DVFS clock):
#   GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (8 x the kernel's cycles; check: /8/duration = the sclk rocm-smi shows);
#   MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs);  effective clock = GRBM_GUI_ACTIVE / 8 / duration
# usage (GPU box): tools/pmc_mfma_util.sh [fp16|bf16] [views]      (the product attention kernel, one operand format per run)
V=${1:-fp16}; VIEWS=${2:-100}
mkdir -p gpurun_out/pmcu
export TMPDIR=/tmp
CMD="python tools/kernel_bench.py --what attnproduct --attn-dtypes $V --views $VIEWS"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS SQ_INSTS_LDS -d gpurun_out/pmcu/p_$V --output-format csv -- $CMD > gpurun_out/pmcu/p_$V.log 2>&1
python - <<PY
import csv, glob, json, collections
f = glob.glob("gpurun_out/pmcu/p_$V/*/*counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "attn_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in acc.items()}
t = glob.glob("gpurun_out/pmcu/p_$V/*/*kernel_trace.csv")[0]
d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(t)) if "attn_kernel" in r["Kernel_Name"]]
dur = sum(d) / len(d)
out = {"kernel": "attn_kernel (product)", "operands": "$V", "views": $VIEWS, "counters_per_dispatch": avg, "avg_dispatch_ns": dur}
if "GRBM_GUI_ACTIVE" in avg and avg["GRBM_GUI_ACTIVE"] > 0:
    cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
    out["kernel_cycles"] = cyc
    out["mfma_util_cycles"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)

    out["effective_clock_ghz"] = cyc / dur
print(json.dumps(out))
open("gpurun_out/pmcu/mfma_util_$V.json", "w").write(json.dumps(out, indent=1))
PY
find gpurun_out/pmcu -name "*kernel_trace.csv" -delete
