#!/bin/bash
# GPU box: rocprofv3 kernel stats of the default (graph) bench run -> gpurun_out/quick_stats.csv + a top-40 table on stdout
export NPP_BENCH_SUPERVISE=0 GPU_MAX_HW_QUEUES=2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/qs
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qs -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/quick_stats.log 2>&1
s=$(find /tmp/qs -name "*kernel_stats.csv" | head -1)
cp "$s" gpurun_out/quick_stats.csv
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/quick_stats.csv")))
# steps covered by the trace: every step (warm-up, capture and replays alike) launches pack_weights_batched_kernel exactly once
steps = next((int(r["Calls"]) for r in rows if "pack_weights_batched_kernel" in r["Name"]), 14)
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
print(f"kernel time per step {tot:.2f} ms, launches per step {sum(int(r['Calls']) for r in rows) / steps:.0f}")
for r in rows[:40]:
    print(f"{r['Name'].replace('void (anonymous namespace)::', '')[:75]:75s} {int(r['Calls']) / steps:7.1f} {float(r['TotalDurationNs']) / steps / 1e6:7.3f} ms {float(r['AverageNs']) / 1e3:8.1f} us")
PY
