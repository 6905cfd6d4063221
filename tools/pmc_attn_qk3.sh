export TMPDIR=/tmp; mkdir -p gpurun_out/r6o; cd /tmp
for planes in 3 2 1; do
  if [ $planes = 1 ]; then CMD="python $OLDPWD/tools/kernel_bench.py --what attnsel --attn-dtypes fp16 --views 100 --sels 2"; else CMD="python $OLDPWD/tools/robust_attn_ab.py --worker --planes $planes --views 100 --iters 3"; fi
  PYTHONPATH=$OLDPWD rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OLDPWD/gpurun_out/r6o/pmc_a_$planes --output-format csv -- $CMD > $OLDPWD/gpurun_out/r6o/a_$planes.log 2>&1
  PYTHONPATH=$OLDPWD rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OLDPWD/gpurun_out/r6o/pmc_b_$planes --output-format csv -- $CMD > $OLDPWD/gpurun_out/r6o/b_$planes.log 2>&1
done
cd $OLDPWD
python - <<'PY'
import csv, glob, collections, json
out = {}
for planes in (3, 2, 1):
    acc = collections.defaultdict(list)
    dur = []
    for ab in "ab":
        for f in glob.glob(f"gpurun_out/r6o/pmc_{ab}_{planes}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "f3r_attn_asm" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(f"gpurun_out/r6o/pmc_{ab}_{planes}/*/*kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                if "f3r_attn_asm" in r["Kernel_Name"]:
                    dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
    v = {k: sum(x) / len(x) for k, x in acc.items()}
    v["avg_dispatch_ms"] = sum(dur) / max(1, len(dur))
    if v.get("GRBM_GUI_ACTIVE"):
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        v["mfma_busy_frac_of_simd_cycles"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024)
        v["effective_clock_ghz"] = cyc / (v["avg_dispatch_ms"] * 1e6)
    for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
        if k in v and v.get("SQ_WAVE_CYCLES"):
            v[k + "_frac_of_wave_cycles"] = v[k] / v["SQ_WAVE_CYCLES"]
    if v.get("SQ_INSTS_MFMA"):
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD"):
            if k in v:
                v[k + "_per_mfma"] = v[k] / v["SQ_INSTS_MFMA"]
    out[{3: "qk3f8", 2: "qk3", 1: "one_product"}[planes]] = v
    print(planes, {k: (round(x, 4) if x < 1000 else int(x)) for k, x in v.items()})
json.dump(out, open("gpurun_out/r6o/attn_qk3_pmc.json", "w"), indent=1)
PY
find gpurun_out/r6o -name "*kernel_trace.csv" -delete; find gpurun_out/r6o -name "*counter_collection.csv" -delete; rm -rf gpurun_out/r6o/pmc_*/*/*.db
