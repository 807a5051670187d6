#!/bin/bash
# GPU box: rocprofv3 kernel stats of the supernet bench run (config 5, weights pass) -> gpurun_out/quick_stats_search.csv + top table
export NPP_BENCH_SUPERVISE=0 GPU_MAX_HW_QUEUES=2 NPP_STREAMS=1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/qss
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qss -- python3 bench.py --model search --batch 8 --steps 4 --warmup 2 --graph 0 --no-cpu-baseline --no-prof > gpurun_out/quick_stats_search.log 2>&1
tail -1 gpurun_out/quick_stats_search.log | cut -c1-200
s=$(find /tmp/qss -name "*kernel_stats.csv" | head -1)
cp "$s" gpurun_out/quick_stats_search.csv
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/quick_stats_search.csv")))
# steps covered by the trace: every step (warm-up, capture and replays alike) launches pack_weights_batched_kernel exactly once
steps = next((int(r["Calls"]) for r in rows if "pack_weights_batched_kernel" in r["Name"]), 6)
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
print(f"kernel time per step ~{tot:.2f} ms, launches per step ~{sum(int(r['Calls']) for r in rows) / steps:.0f}")
for r in rows[:45]:
    print(f"{r['Name'].replace('void (anonymous namespace)::', '')[:80]:80s} {int(r['Calls']) / steps:7.1f} {float(r['TotalDurationNs']) / steps / 1e6:7.3f} ms {float(r['AverageNs']) / 1e3:8.1f} us")
PY
