"""Idle time between consecutive kernels of one HIP queue, from a rocprofv3 --kernel-trace CSV of the hipGraph bench run:
how much of the step is launch gaps rather than kernels.  usage: gap_stats.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * skip):]            # the replayed steps at the end of the run
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
print(f"window {(t1 - t0) / 1e6:.2f} ms, {len(rows)} launches")
byq = defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", "?")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
# union of busy intervals over all queues: time with at least one / at least two kernels running
ev = []
for q, ks in byq.items():
    for s, e, _ in ks:
        ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, busy = 0, t0, defaultdict(int)
for t, d in ev:
    busy[min(depth, 3)] += t - last
    last, depth = t, depth + d
tot = t1 - t0
print("time with 0 / 1 / 2 / >=3 kernels in flight: " + " / ".join(f"{busy[k] / tot:.1%}" for k in range(4)))
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    ks.sort()
    kt = sum(e - s for s, e, _ in ks)
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    pos = [g for g in gaps if g > 0]
    small = [g for g in pos if g < 20000]
    print(f"queue {q}: {len(ks)} launches, kernel time {kt / 1e6:.2f} ms, positive gaps {sum(pos) / 1e6:.2f} ms "
          f"(median {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.2f} us, <20us gaps: {len(small)} totalling {sum(small) / 1e6:.2f} ms)")
