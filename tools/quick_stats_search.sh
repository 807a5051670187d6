#!/bin/bash
export NPP_BENCH_SUPERVISE=0   # under rocprofv3 the profiled process must be the worker itself: never a supervisor that spawns one
# GPU box: rocprofv3 --kernel-trace --stats of the search-supernet bench (config 5) -> gpurun_out/quick_stats_search.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/qss; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qss -- python3 bench.py --model search --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --no-prof > /tmp/qss.log 2>&1
grep -h "^{" /tmp/qss.log | tail -1 | cut -c1-160
s=$(find /tmp/qss -name "*kernel_stats.csv" | head -1)
python3 - "$s" <<'PY' > gpurun_out/quick_stats_search.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 1
for r in rows:
    if "pack_weights_batched" in r["Name"]:
        steps = int(r["Calls"])
tot = sum(int(r["TotalDurationNs"]) for r in rows) / steps / 1e6
n = sum(int(r["Calls"]) for r in rows) / steps
print("steps", steps, "total kernel ms/step", round(tot, 2), "launches/step", round(n))
cum = 0
for r in rows[:45]:
    ms = int(r["TotalDurationNs"]) / steps / 1e6
    cum += ms
    nm = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{ms:7.2f} {cum:7.2f} {int(r['Calls']) // steps:5d} {float(r['AverageNs']) / 1e3:8.1f}us  {nm[:110]}")
PY
