#!/bin/bash
# PMC evidence for the fusion-attention kernel at T = 327 680 (N = 320), both operand formats, kernel_sel as given (default 2 =
# the hand-scheduled kernel): matrix-pipe utilisation in cycles + effective clock (one pass), FETCH_SIZE and WRITE_SIZE (separate passes,
# MI355X_MICROARCH.md "HBM").  Writes gpurun_out/pmc_r03/attn_mfma_util.json and gpurun_out/pmc_r03/attn_traffic_new.json.
SEL=${1:-2}; VIEWS=${2:-320}
out=gpurun_out/pmc_attn; mkdir -p $out
export TMPDIR=/tmp
for V in fp16 bf16; do
  CMD="python tools/kernel_bench.py --what attnsel --attn-dtypes $V --views $VIEWS --sels $SEL"
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $out/u_$V --output-format csv -- $CMD > $out/u_$V.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/f_$V --output-format csv -- $CMD > $out/f_$V.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/w_$V --output-format csv -- $CMD > $out/w_$V.log 2>&1
done
python - <<PY
import csv, glob, json, collections
out = "$out"
res = {"_doc": "rocprofv3 --kernel-trace --pmc (tools/pmc_attn_util.sh) over tools/kernel_bench.py --what attnsel --sels $SEL at T = %d (N = $VIEWS): the fusion-attention kernel f3r_attn_fwd takes for that shape (kernel_sel $SEL).  mfma_util_cycles = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); mfma_util_useful_cycles = the algorithm's 32x32x16 MFMA instructions x 32 pipe cycles over the same denominator (layout 2 of the hand-scheduled kernel issues no other MFMA, so the two agree; layout 1 and the HIP kernel spend extra pipe cycles on bias steps); effective clock = GRBM_GUI_ACTIVE/8/duration.  Profiled runs clock a few per cent below plain ones." % ($VIEWS * 1024), "formats": {}}
traffic = {}
for V in ("fp16", "bf16"):
    def counters(tag):
        f = glob.glob(f"{out}/{tag}_{V}/*/*counter_collection.csv")[0]
        acc = collections.defaultdict(list)
        names = set()
        for r in csv.DictReader(open(f)):
            if "attn" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                names.add(r["Kernel_Name"])
        return {k: sum(v) / len(v) for k, v in acc.items()}, sorted(names)
    avg, names = counters("u")
    t = glob.glob(f"{out}/u_{V}/*/*kernel_trace.csv")[0]
    d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(t)) if "attn" in r["Kernel_Name"]]
    dur = sum(d) / len(d)
    cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
    e = {"views": $VIEWS, "kernel": names, "avg_dispatch_ms": dur / 1e6, "kernel_cycles": cyc,
         "mfma_util_cycles": avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
         "effective_clock_ghz": cyc / dur, "valu_insts_per_mfma": avg["SQ_INSTS_VALU"] / avg["SQ_INSTS_MFMA"],
         "lds_insts_per_mfma": avg.get("SQ_INSTS_LDS", 0) / avg["SQ_INSTS_MFMA"], "counters_per_dispatch": avg}
    T = $VIEWS * 1024
    big = 4.0 * T * T * 64 * 16 / (2 * 32 * 32 * 16)   # wave-level 32x32x16 MFMA instructions of the algorithm
    e["mfma_util_useful_cycles"] = big * 32 / (cyc * 1024)
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in avg and avg.get("SQ_WAVE_CYCLES"):
            e[k + "_frac_of_wave_cycles"] = avg[k] / avg["SQ_WAVE_CYCLES"]
    e["tflops_profiled"] = 4.0 * T * T * 64 * 16 / dur / 1e3
    res["formats"][V] = e
    f, _ = counters("f")
    w, _ = counters("w")
    traffic[V] = {"FETCH_SIZE_KiB": f["FETCH_SIZE"], "WRITE_SIZE_KiB": w["WRITE_SIZE"], "bytes": (2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024, "kernel": names}
json.dump(res, open(f"{out}/attn_mfma_util.json", "w"), indent=1)
json.dump(traffic, open(f"{out}/attn_traffic_new.json", "w"), indent=1)
print(json.dumps({V: {k: res["formats"][V][k] for k in ("avg_dispatch_ms", "mfma_util_cycles", "mfma_util_useful_cycles", "effective_clock_ghz", "tflops_profiled")} for V in res["formats"]}))
print(json.dumps(traffic))
PY
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete; rm -rf $out/*/*/*.db 2>/dev/null
