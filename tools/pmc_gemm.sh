#!/bin/bash
# rocprofv3 PMC over the GEMM micro-benchmark (M = 40960, the four ViT-L linear shapes + QKV): matrix-pipe utilisation in cycles and the
# LDS / VMEM instruction mix per gemm_kernel / gemm256_kernel instantiation and grid.  Two --pmc passes (own runs, kernel-trace only).
mkdir -p gpurun_out/pmcg
export TMPDIR=/tmp
CMD="python tools/kernel_bench.py --what gemm"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d gpurun_out/pmcg/p1 --output-format csv -- $CMD > gpurun_out/pmcg/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE -d gpurun_out/pmcg/p2 --output-format csv -- $CMD > gpurun_out/pmcg/p2.log 2>&1
python - <<'PY'
import csv, glob, json, collections
out = collections.defaultdict(dict)
for p in ("p1", "p2"):
    fs = glob.glob(f"gpurun_out/pmcg/{p}/*/*counter_collection.csv")
    if not fs:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "gemm_kernel" in r["Kernel_Name"] or "gemm256_kernel" in r["Kernel_Name"]:
            key = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(f3r_gemm_args")[0] + f" grid={r.get('Grid_Size', '?')}"
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        for c, v in d.items():
            out[k][c] = sum(v) / len(v)
for k, d in out.items():
    if "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"] > 0:
        cyc = d["GRBM_GUI_ACTIVE"] / 8.0
        d["mfma_util_cycles"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024)
        if d.get("SQ_INSTS_MFMA"):
            d["lds_insts_per_mfma"] = d.get("SQ_INSTS_LDS", 0.0) / d["SQ_INSTS_MFMA"]
            d["vmem_rd_per_mfma"] = d.get("SQ_INSTS_VMEM_RD", 0.0) / d["SQ_INSTS_MFMA"]
json.dump(out, open("gpurun_out/pmcg/gemm_pmc.json", "w"), indent=1)
for k, d in out.items():
    print(k, {c: (round(v, 4) if v < 100 else int(v)) for c, v in d.items() if c in ("mfma_util_cycles", "lds_insts_per_mfma", "vmem_rd_per_mfma", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_MFMA")})
PY
grep tflops gpurun_out/pmcg/p1.log | cut -c1-200
find gpurun_out/pmcg -name "*kernel_trace.csv" -delete
