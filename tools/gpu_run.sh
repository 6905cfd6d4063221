#!/bin/bash
# One parametrised script for every gpurun call of a round (replaces the one-shot tools/gpu_r2_*.sh / gpu_r3_*.sh of earlier rounds):
#     gpurun -- 'bash tools/gpu_run.sh STEP [STEP ...]'
# Each step writes its logs under gpurun_out/<step>/; summaries worth judging are copied to profiles/ by hand afterwards.
export TMPDIR=/tmp
mkdir -p gpurun_out
for step in "$@"; do
  d=gpurun_out/$step; mkdir -p $d
  t0=$(date +%s)
  case $step in
    depth)      # parity at depth + GEMM roles at N = 320 shapes (tests/test_depth_parity_gpu.py)
      timeout 1500 python -m pytest tests/test_depth_parity_gpu.py -q -rA -s -p no:cacheprovider 2>&1 | tail -120 > $d/pytest.log; grep -E "parity|passed|failed|FAILED|Error" $d/pytest.log | tail -40 ;;
    gemmasm)    # parity of the hand-scheduled GEMM kernels + the role tests at N = 320 shapes (automatic selection takes them there)
      timeout 900 python -m pytest tests/test_gemm_asm_gpu.py tests/test_depth_parity_gpu.py -q -rA -p no:cacheprovider -k "gemm_asm or role" 2>&1 | tail -60 > $d/pytest.log; grep -E "passed|failed|FAILED|Error|error" $d/pytest.log | tail -30 ;;
    gemmlab)    # variants of the hand-scheduled GEMM (tools/gemm_lab.py --build on the CPU box first) + a PMC pass of the product variant
      timeout 600 python tools/gemm_lab.py --run ${GEMMLAB_ARGS:-} > $d/gemm_lab.jsonl 2> $d/err.log; cat $d/gemm_lab.jsonl | cut -c1-250; tail -3 $d/err.log
      if [ -z "${NO_PMC:-}" ]; then
      ( cd /tmp; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $OLDPWD/$d/pmc --output-format csv -- python $OLDPWD/tools/gemm_lab.py --run --variants product --rounds 2 --shapes ${PMC_SHAPES:-lp:327680:4096:4096,lp:327680:4096:1024,f32:327680:1024:4096} > $OLDPWD/$d/pmc.log 2>&1 )
      python - $d <<'PY'
import csv, glob, sys, collections, json
d = sys.argv[1]
fs = glob.glob(f"{d}/pmc/*/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in fs:
    for r in csv.DictReader(open(f)):
        if "f3r_gemm_asm" in r["Kernel_Name"]:
            acc[r["Kernel_Name"] + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    v = {n: sum(x) / len(x) for n, x in c.items()}
    if v.get("GRBM_GUI_ACTIVE"):
        v["mfma_util_cycles"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024)
    out[k] = v
    print(k, {n: (round(x, 4) if x < 100 else int(x)) for n, x in v.items()})
json.dump(out, open(f"{d}/gemm_asm_pmc.json", "w"), indent=1)
PY
      find $d/pmc -name "*kernel_trace.csv" -delete
      fi ;;
    convf8pmc)  # the DPT head's conv roles, split x3 vs x3f8: event timings, then two rocprofv3 --pmc passes over the same launches (matrix-pipe busy, waits, instruction mix)
      timeout 600 python tools/conv_f8_ab.py > $d/conv_roles.jsonl 2> $d/err.log; cat $d/conv_roles.jsonl | cut -c1-200
      ( cd /tmp; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $OLDPWD/$d/pmc1 --output-format csv -- python $OLDPWD/tools/conv_f8_ab.py --roles head2,head0,rcu128 --reps 2 > $OLDPWD/$d/pmc1.log 2>&1
        rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS -d $OLDPWD/$d/pmc2 --output-format csv -- python $OLDPWD/tools/conv_f8_ab.py --roles head2,head0,rcu128 --reps 2 > $OLDPWD/$d/pmc2.log 2>&1 )
      python - $d <<'PY'
import csv, glob, sys, collections, json
d = sys.argv[1]
out = {"source": "rocprofv3 --kernel-trace --pmc (two passes) over tools/conv_f8_ab.py --roles head2,head0,rcu128 --reps 2; averages per launch",
       "derived": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024); waits as fractions of SQ_WAVE_CYCLES", "kernels": {}}
for sub in ("pmc1", "pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{d}/{sub}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "gemm256" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("::")[-1][:70] + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        out["kernels"].setdefault(k, {}).update({n: sum(x) / len(x) for n, x in c.items()})
for k, v in out["kernels"].items():
    if v.get("GRBM_GUI_ACTIVE"):
        v["mfma_util"] = round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024), 4)
    if v.get("SQ_WAVE_CYCLES"):
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if n in v:
                v[n + "_frac"] = round(v[n] / v["SQ_WAVE_CYCLES"], 4)
    print(k, {n: (round(x, 4) if x < 100 else int(x)) for n, x in v.items()})
json.dump(out, open(f"{d}/conv_x3_vs_x3f8_pmc.json", "w"), indent=1)
PY
      find $d/pmc1 $d/pmc2 -name "*kernel_trace.csv" -delete; rm -rf $d/pmc1 $d/pmc2 ;;
    attnhd)     # the generated head_dim-80 / 128 attention kernels: parity tests, then TF/s beside the generic HIP kernel and the head_dim-64 kernel
      timeout 900 python -m pytest tests/test_attn_asm_gpu.py tests/test_kernels_gpu.py -q -rA -p no:cacheprovider -k "head_dim or other_head" 2>&1 | tail -150 > $d/pytest.log; grep -E "passed|failed|FAILED|Error" $d/pytest.log | tail -20
      timeout 600 python tools/kernel_bench.py --what attnhd --views ${ATTNHD_VIEWS:-100} --attn-dtypes ${ATTNHD_DTYPES:-fp16,bf16} > $d/attn_head_dim.jsonl 2> $d/err.log; cat $d/attn_head_dim.jsonl | cut -c1-330; tail -3 $d/err.log ;;
    attnhdpmc)  # matrix-pipe utilisation + effective clock of the generated attention kernels at head_dim 64 / 80 / 128 (one PMC pass, T = 102 400)
      ( cd /tmp; PYTHONPATH=$OLDPWD rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $OLDPWD/$d/pmc --output-format csv -- python $OLDPWD/tools/kernel_bench.py --what attnhd --views 100 --attn-dtypes ${ATTNHD_DTYPES:-fp16} ) > $d/pmc.log 2>&1
      python - $d <<'PY'
import csv, glob, sys, collections, json
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{d}/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(f"{d}/pmc/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:60]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for k, c in acc.items():
    v = {n: sum(x) / len(x) for n, x in c.items()}
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    t = sum(dur[k]) / max(1, len(dur[k]))
    out[k] = {"avg_dispatch_ms": t / 1e6, "mfma_util_cycles": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), "effective_clock_ghz": cyc / t if t else None,
              "valu_insts_per_mfma": v["SQ_INSTS_VALU"] / max(1.0, v["SQ_INSTS_MFMA"]), "lds_insts_per_mfma": v["SQ_INSTS_LDS"] / max(1.0, v["SQ_INSTS_MFMA"]),
              "wait_inst_frac_of_wave_cycles": v.get("SQ_WAIT_INST_ANY", 0) / max(1.0, v.get("SQ_WAVE_CYCLES", 1)), "mfma_insts": v["SQ_INSTS_MFMA"]}
    print(k, {n: (round(x, 4) if isinstance(x, float) else x) for n, x in out[k].items()})
json.dump(out, open(f"{d}/attn_head_dim_pmc.json", "w"), indent=1)
PY
      find $d/pmc -name "*kernel_trace.csv" -delete; find $d/pmc -name "*counter_collection.csv" -delete; rm -rf $d/pmc/*/*.db 2>/dev/null ;;
    smalln)     # small scenes (fp16 / high): latency eager vs graph, then the kernel split of N = ${SMALLN_PROFILE_VIEWS:-3} eager forwards
      timeout 600 python tools/small_n_latency.py --dtype fp16 --precision high --views ${SMALLN_VIEWS:-2,3,8,20} > $d/latency.jsonl 2> $d/err.log; cat $d/latency.jsonl; tail -2 $d/err.log
      ( cd /tmp; PYTHONPATH=$OLDPWD rocprofv3 --kernel-trace --stats -d $OLDPWD/$d/prof --output-format csv -- python $OLDPWD/tools/small_n_latency.py --dtype fp16 --precision high --views ${SMALLN_PROFILE_VIEWS:-3} --no-graph --iters 10 ) > $d/prof.log 2>&1
      f=$(ls $d/prof/*/*kernel_stats.csv | head -1); cp $f $d/kernel_stats.csv; head -40 $d/kernel_stats.csv | cut -c1-200
      find $d/prof -name "*kernel_trace.csv" -delete; rm -rf $d/prof/*/*.db 2>/dev/null ;;
    huge)       # the reference's model_scaling_huge decoder (head_dim 80) end to end: generated kernel vs the generic one
      for f in "" "--generic"; do timeout 600 python tools/huge_decoder_bench.py --views ${HUGE_VIEWS:-100} $f 2>> $d/err.log | tee -a $d/huge_decoder.jsonl | cut -c1-600; done; tail -2 $d/err.log ;;
    attnl2)     # L2 hit / miss / fabric read requests of the fusion-attention kernel at N = 320, fp16 vs bf16 (why fp16 fetches 2-3x the tiling floor)
      rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u | tr "\n" " " > $d/tcc_counters.txt
      for V in fp16 bf16; do
        ( cd /tmp; PYTHONPATH=$OLDPWD rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $OLDPWD/$d/p_$V --output-format csv -- python $OLDPWD/tools/kernel_bench.py --what attnsel --attn-dtypes $V --views 320 --sels 2 > $OLDPWD/$d/p_$V.log 2>&1 )
      done
      python - $d <<'PY'
import csv, glob, sys, json, collections
d = sys.argv[1]
out = {}
for V in ("fp16", "bf16"):
    fs = glob.glob(f"{d}/p_{V}/*/*counter_collection.csv")
    acc = collections.defaultdict(list)
    per_disp = collections.defaultdict(dict)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if "attn" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                per_disp[r.get("Dispatch_Id", "?")][r["Counter_Name"]] = float(r["Counter_Value"])
    v = {k: sum(x) / len(x) for k, x in acc.items()}
    if v.get("TCC_HIT_sum") is not None and v.get("TCC_MISS_sum") is not None:
        v["l2_hit_rate"] = v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
    v["per_dispatch_miss"] = [int(x.get("TCC_MISS_sum", -1)) for x in per_disp.values()]
    out[V] = v
    print(V, {k: (round(x, 4) if isinstance(x, float) and x < 10 else x) for k, x in v.items()})
json.dump(out, open(f"{d}/attn_l2.json", "w"), indent=1)
PY
      find $d -name "*kernel_trace.csv" -delete ;;
    attnpmc)    # matrix-pipe utilisation + effective clock + FETCH / WRITE of the fusion attention at N = 320, both formats (tools/pmc_attn_util.sh)
      timeout 900 bash tools/pmc_attn_util.sh 2 320 > $d/pmc.log 2>&1; tail -3 $d/pmc.log | cut -c1-600; cp gpurun_out/pmc_attn/attn_mfma_util.json gpurun_out/pmc_attn/attn_traffic_new.json $d/ 2>/dev/null ;;
    ubench8)    # go / no-go for block-scaled fp8 / fp6 correction planes: the matrix pipe under the power cap on mixed fp16 + f8f6f4 streams
      timeout 300 tools/ubench/mfma_mixed ${UBENCH_SECONDS:-1.0} > $d/mfma_mixed.jsonl 2> $d/err.log; cat $d/mfma_mixed.jsonl | cut -c1-400; tail -3 $d/err.log ;;
    convpmc)    # matrix-pipe utilisation + instruction mix of the conv / QKV instantiations of the compiler-scheduled kernels (no PMC pass of them existed)
      ( cd /tmp; PYTHONPATH=$OLDPWD rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $OLDPWD/$d/pmc --output-format csv -- python $OLDPWD/tools/kernel_bench.py --what convheads > $OLDPWD/$d/pmc.log 2>&1 )
      grep tflops $d/pmc.log | cut -c1-300
      python - $d <<'PY'
import csv, glob, sys, collections, json
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{d}/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            key = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(f3r_gemm_args")[0] + f" grid={r.get('Grid_Size', '?')}"
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(f"{d}/pmc/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            key = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(f3r_gemm_args")[0] + f" grid={r.get('Grid_Size', '?')}"
            dur[key].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for k, c in acc.items():
    v = {n: sum(x) / len(x) for n, x in c.items()}
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8.0
    t = sum(dur[k]) / max(1, len(dur[k]))
    if cyc:
        v["mfma_util_cycles"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024)
        v["effective_clock_ghz"] = cyc / t if t else None
        v["avg_dispatch_ms"] = t / 1e6
        if v.get("SQ_INSTS_MFMA"):
            v["lds_insts_per_mfma"] = v.get("SQ_INSTS_LDS", 0) / v["SQ_INSTS_MFMA"]
            v["valu_insts_per_mfma"] = v.get("SQ_INSTS_VALU", 0) / v["SQ_INSTS_MFMA"]
    out[k] = v
    print(k, {n: (round(x, 4) if x < 100 else int(x)) for n, x in v.items() if n in ("mfma_util_cycles", "effective_clock_ghz", "avg_dispatch_ms", "lds_insts_per_mfma", "valu_insts_per_mfma")})
json.dump(out, open(f"{d}/conv_pmc.json", "w"), indent=1)
PY
      find $d/pmc -name "*kernel_trace.csv" -delete ;;
    benchquick) # the default line, short: live roofline + power sampler + n100 / fusion_only_n20 objects (no alt format, no CPU baseline, no hot weights)
      timeout 900 python bench.py --steps ${BENCH_STEPS:-1} --warmup 1 --no-alt --no-cpu-baseline --no-hot ${BENCH_EXTRA:-} > $d/bench_quick.json 2> $d/err.log; tail -c 5000 $d/bench_quick.json; tail -5 $d/err.log ;;
    attnasm)    # the generated attention kernels (the clock bracket of round 5 touched prologue + epilogue)
      timeout 900 python -m pytest tests/test_attn_asm_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -40 > $d/pytest.log; tail -12 $d/pytest.log ;;
    gemmf8)     # parity of the fp8-low-plane GEMM + LayerNorm rows, then W2 vs W2F8 timings of the transformer's linear roles
      timeout 900 python -m pytest tests/test_gemm_asm_gpu.py -q -x -s -p no:cacheprovider -k "fp8 or f8" 2>&1 | tail -40 > $d/pytest.log; grep -E "w2f8|passed|failed|Error|assert" $d/pytest.log | tail -20
      timeout 600 python tools/kernel_bench.py --what gemmf8 --views ${F8_VIEWS:-320,100} > $d/gemm_w2_vs_w2f8.jsonl 2> $d/err.log; cat $d/gemm_w2_vs_w2f8.jsonl | cut -c1-300; tail -3 $d/err.log ;;
    f8model)    # the model with Fast3R.low_plane = "fp8": kernel chain test, parity at ViT-L depth (N = 3, 8 vs the CPU oracle; N = 100 vs the exact mode), then the bench A/B
      timeout 1500 python -m pytest tests/test_gemm_asm_gpu.py tests/test_depth_parity_gpu.py -q -s -p no:cacheprovider -k "fp8 or f8 or stress_weights_vs_cpu_oracle or n100_stress" > $d/pytest_full.log 2>&1; grep -E "\[parity\]|\[w2f8|passed|failed|^E " $d/pytest_full.log | sed 's/^[.sF]*//' | cut -c1-300 > $d/parity_lines.txt; cat $d/parity_lines.txt; tail -60 $d/pytest_full.log > $d/pytest.log; rm -f $d/pytest_full.log
      for lpn in fp8 fp16; do timeout 600 python bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline --no-hot --no-inference --low-plane $lpn > $d/bench_$lpn.json 2> $d/err_$lpn.log; python - $d/bench_$lpn.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = lambda x, n=2: None if x is None else round(x, n)
print(sys.argv[1], "N=320:", r(d.get("value")), "views/s", r(d.get("ms_per_step"), 1), "ms; attn frac", r(d.get("roofline", {}).get("frac"), 4), "| n100:", r(d.get("n100", {}).get("value")),
      "views/s | n20 fusion:", r(d.get("fusion_only_n20", {}).get("value"), 1), "| error:", d.get("error"), d.get("n100", {}).get("error"))
PY
      done ;;
    emurank)    # EMULATION: one GPU runs rank 3 of 8 of the view-sharded forward at N = 320 (and N = 1500) with the round's kernels (no collectives)
      timeout 900 python bench.py --emulate-rank 3 --of 8 --steps 2 --warmup 1 > $d/emulated_rank3of8_n320.json 2> $d/err.log; python -c "
import json; d = json.load(open('$d/emulated_rank3of8_n320.json')); print('rank 3 of 8, N=320:', round(d['per_rank_step_ms'], 1), 'ms ->', round(d['projected_views_per_s_if_comm_is_hidden'], 1), 'views/s if the exchange is hidden')" ;;
    gemmf8pmc)  # matrix-pipe utilisation + effective clock of the hand-scheduled GEMM kernels, two fp16 planes vs the low plane in fp8 (one PMC pass, M = 327 680)
      ( cd /tmp; PYTHONPATH=$OLDPWD rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $OLDPWD/$d/pmc --output-format csv -- python $OLDPWD/tools/kernel_bench.py --what gemmf8 --views 320 ) > $d/pmc.log 2>&1
      python - $d <<'PY'
import csv, glob, sys, collections, json
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{d}/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "f3r_gemm_asm" in r["Kernel_Name"]:
            acc[r["Kernel_Name"] + " grid=" + r.get("Grid_Size", "?") + " lds=" + r.get("LDS_Block_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(f"{d}/pmc/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "f3r_gemm_asm" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for k, c in acc.items():
    v = {n: sum(x) / len(x) for n, x in c.items()}
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if cyc:
        v["mfma_util_cycles"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024)
        v["launches"] = len(next(iter(c.values())))
    out[k] = v
    print(k, {n: (round(x, 4) if x < 100 else int(x)) for n, x in v.items() if n in ("mfma_util_cycles", "SQ_INSTS_MFMA", "launches")})
for k, x in dur.items():
    print(k, "mean dispatch ms over all roles", round(sum(x) / len(x) / 1e6, 3), "n", len(x))
json.dump(out, open(f"{d}/gemm_asm_f8_pmc.json", "w"), indent=1)
PY
      find $d/pmc -name "*kernel_trace.csv" -delete ;;
    qkvasm)     # QKV on the hand-scheduled kernel: RoPE-2D epilogue + fp8 low plane (tests), then the model parity at depth and the bench A/B vs low_plane fp16
      timeout 900 python -m pytest tests/test_gemm_asm_gpu.py -q -x -s -p no:cacheprovider -k "qkv" 2>&1 | tail -30 > $d/pytest.log; grep -E "passed|failed|Error|assert|^E " $d/pytest.log | tail -12
      timeout 900 python -m pytest tests/test_depth_parity_gpu.py tests/test_e2e_gpu.py -q -s -p no:cacheprovider -k "stress_weights_vs_cpu_oracle or n100_stress or vit_large_512_n8 or sharded_forward_equals_unsharded" > $d/pytest_full.log 2>&1; grep -E "\[parity\]|passed|failed|^E " $d/pytest_full.log | sed 's/^[.sF]*//' | cut -c1-260 > $d/parity_lines.txt; cat $d/parity_lines.txt; tail -40 $d/pytest_full.log > $d/pytest2.log; rm -f $d/pytest_full.log
      timeout 600 python bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline --no-hot --no-inference > $d/bench_fp8.json 2> $d/err.log; python - $d/bench_fp8.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = lambda x, n=2: None if x is None else round(x, n)
print(sys.argv[1], "N=320:", r(d.get("value")), "views/s", r(d.get("ms_per_step"), 1), "ms; attn frac", r(d.get("roofline", {}).get("frac"), 4), "| n100:", r(d.get("n100", {}).get("value")),
      "views/s | n20 fusion:", r(d.get("fusion_only_n20", {}).get("value"), 1), "| error:", d.get("error"), d.get("n100", {}).get("error"))
PY
      ;;
    steal)      # the fusion attention with and without work stealing, interleaved rounds, N = 320 and N = 100 (tools/kernel_bench.py --what attnsteal)
      timeout 600 python tools/kernel_bench.py --what attnsteal --views ${STEAL_VIEWS:-320,100,20} > $d/attn_work_stealing.jsonl 2> $d/err.log; cat $d/attn_work_stealing.jsonl | cut -c1-400; tail -3 $d/err.log ;;
    gpuslow)    # the GPU tests kept out of -m gpu (conftest.py: gpu_slow)
      timeout 1800 python -m pytest tests -m gpu_slow -q -rA -s -p no:cacheprovider 2>&1 | tail -40 > $d/pytest.log; grep -E "passed|failed|\[parity" $d/pytest.log | tail -12 ;;
    two)        # selected tests, verbose
      timeout 900 python -m pytest ${TWO_TESTS} -q -x -s -rA -p no:cacheprovider > $d/pytest.log 2>&1; grep -E "\[parity\]|\[exact|passed|failed|Error|assert" $d/pytest.log | cut -c1-400 | tail -30 ;;
    gputests)   # the whole GPU suite + smoke
      timeout 2400 python -m pytest tests -m gpu -q -rA -s --durations=70 -p no:cacheprovider > $d/pytest_full.log 2>&1; grep -E "^\[parity\]|^\[exact" $d/pytest_full.log > $d/parity_lines.txt; sed -n '/slowest/,/short test summary/p' $d/pytest_full.log > $d/durations.txt; grep -E "passed|failed" $d/pytest_full.log | tail -3; grep -E "^(FAILED|ERROR)" $d/pytest_full.log | head -20; sed -n '/=== FAILURES ===/,/=== short test summary/p' $d/pytest_full.log | tail -400 > $d/failures.txt; grep -E "\[parity\]|\[exact" $d/pytest_full.log | sed 's/^[.sF]*//' > $d/parity_lines.txt; tail -150 $d/pytest_full.log > $d/pytest.log; rm -f $d/pytest_full.log
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $d/smoke.log 2>&1; tail -2 $d/smoke.log ;;
    hot100)     # N = 100 on the hot weights: views/s, attention frac, re-base rate, parity vs the exact mode -- and the default weights beside it
      timeout 900 python bench.py --views 100 --weights hot --steps 3 --warmup 1 --no-alt --no-cpu-baseline --parity-exact > $d/bench_n100_hot.json 2> $d/hot.err; tail -c 1500 $d/bench_n100_hot.json
      timeout 900 python bench.py --views 100 --steps 3 --warmup 1 --no-alt --no-cpu-baseline --no-hot > $d/bench_n100_default.json 2> $d/def.err; tail -c 600 $d/bench_n100_default.json ;;
    bench320)   # the default line (N = 320, fp16 / high) with its hot-weights object, short
      timeout 1200 python bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline > $d/bench_n320.json 2> $d/err.log; tail -c 2500 $d/bench_n320.json ;;
    benchfull)  # what the driver runs
      timeout 1500 python bench.py > $d/bench_default.json 2> $d/err.log; tail -c 3000 $d/bench_default.json ;;
    gemmref)    # the transformer GEMM roles at N = 320 beside the vendor library
      timeout 900 python tools/kernel_bench.py --what gemmref --views ${GEMM_VIEWS:-320} --sels ${GEMM_SELS:-2} > $d/gemmref.jsonl 2> $d/err.log; cat $d/gemmref.jsonl | cut -c1-260 ;;
    gemmpmc)    # rocprofv3 PMC over the GEMM micro-benchmark
      timeout 900 bash tools/pmc_gemm.sh > $d/pmc.log 2>&1; tail -30 $d/pmc.log; cp gpurun_out/pmcg/gemm_pmc.json $d/ 2>/dev/null ;;
    prof320)    # rocprofv3 kernel stats of the default bench command (3 forwards)
      ( cd /tmp; rocprofv3 --kernel-trace --stats -d $OLDPWD/$d/prof --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline --no-hot --no-parity --no-inference --no-extra-configs > $OLDPWD/$d/bench.json 2> $OLDPWD/$d/err.log )
      f=$(ls $d/prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $d/kernel_stats.csv && head -12 $d/kernel_stats.csv | cut -c1-200
      find $d/prof -name "*kernel_trace.csv" -delete ;;
    *) echo "unknown step $step" ;;
  esac
  echo "== $step: $(( $(date +%s) - t0 )) s"
done
