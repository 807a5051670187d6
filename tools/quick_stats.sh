#!/bin/bash
export NPP_BENCH_SUPERVISE=0   # under rocprofv3 the profiled process must be the worker itself: never a supervisor that spawns one
# GPU box: rocprofv3 --kernel-trace --stats of the default bench (hipGraph, two streams) -> gpurun_out/quick_stats.txt (per-step table)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/qs; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qs -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > /tmp/qs.log 2>&1
tail -1 /tmp/qs.log | cut -c1-160
s=$(find /tmp/qs -name "*kernel_stats.csv" | head -1)
python3 - "$s" <<'PY' > gpurun_out/quick_stats${1}.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 1
for r in rows:
    if "pack_weights_batched" in r["Name"]:
        steps = int(r["Calls"])
tot = sum(int(r["TotalDurationNs"]) for r in rows) / steps / 1e6
print("steps", steps, "total kernel ms/step", round(tot, 2))
cum = 0
for r in rows[:70]:
    ms = int(r["TotalDurationNs"]) / steps / 1e6
    cum += ms
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{ms:7.2f} {cum:7.2f} {int(r['Calls']) // steps:5d} {float(r['AverageNs']) / 1e3:8.1f}us  {n[:110]}")
PY
