#!/bin/bash
# rocprofv3 PMC of the attention kernel a library takes for the fusion shape (kernel_sel given): matrix-pipe utilisation in cycles,
# effective clock, wave-cycle buckets.  usage (GPU box): tools/pmc_attn_asm.sh TAG [fp16|bf16] [views] [kernel_sel] ; F3R_LAB_LIB selects a lab library
TAG=${1:-product}; V=${2:-fp16}; VIEWS=${3:-100}; SEL=${4:-2}
mkdir -p gpurun_out/pmca
export TMPDIR=/tmp
CMD="python tools/kernel_bench.py --what attnsel --attn-dtypes $V --views $VIEWS --sels $SEL"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES -d gpurun_out/pmca/p_$TAG --output-format csv -- $CMD > gpurun_out/pmca/p_$TAG.log 2>&1
python - <<PY
import csv, glob, json, collections
f = glob.glob("gpurun_out/pmca/p_$TAG/*/*counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "attn" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in acc.items()}
t = glob.glob("gpurun_out/pmca/p_$TAG/*/*kernel_trace.csv")[0]
d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(t)) if "attn" in r["Kernel_Name"]]
dur = sum(d) / len(d)
out = {"tag": "$TAG", "operands": "$V", "views": $VIEWS, "kernel_sel": $SEL, "counters_per_dispatch": avg, "avg_dispatch_ns": dur}
if avg.get("GRBM_GUI_ACTIVE", 0) > 0:
    cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
    out["kernel_cycles"] = cyc
    out["mfma_util_cycles"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)
    out["effective_clock_ghz"] = cyc / dur
    T = $VIEWS * 1024
    out["tflops_profiled"] = 4.0 * T * T * 64 * 16 / dur / 1e3
    if avg.get("SQ_WAVE_CYCLES", 0) > 0:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in avg:
                out[k + "_frac"] = avg[k] / avg["SQ_WAVE_CYCLES"]
    if avg.get("SQ_INSTS_MFMA", 0) > 0 and "SQ_INSTS_VALU" in avg:
        out["valu_per_mfma"] = avg["SQ_INSTS_VALU"] / avg["SQ_INSTS_MFMA"]
print(json.dumps(out))
open("gpurun_out/pmca/attn_$TAG.json", "w").write(json.dumps(out, indent=1))
PY
find gpurun_out/pmca -name "*kernel_trace.csv" -delete; find gpurun_out/pmca -name "*counter_collection.csv" -delete
