#!/bin/bash
# round 3, GPU call W: where a small scene (N = 4) spends its time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3w; mkdir -p $O; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python tools/small_n_latency.py --views 4 --iters 20 --no-graph > $O/run.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_n4.csv
t=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last forward: take the last ~1/23 of the kernels (3 warm-up + 20 timed forwards)
n = len(rows) // 23
last = rows[-n:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
span = int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])
print("kernels per forward", n, "span ms", span / 1e6, "busy ms", busy / 1e6)
g = collections.defaultdict(lambda: [0, 0])
for r in last:
    k = r["Kernel_Name"].split("(")[0][:70]
    g[k][0] += 1; g[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in sorted(g.items(), key=lambda x: -x[1][1])[:18]:
    print("%-72s %4d %8.3f ms" % (k, v[0], v[1] / 1e6))
PY
cat $O/run.log | tail -2
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/prof
