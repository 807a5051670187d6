"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel count / total / average duration, sorted by total."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
agg = defaultdict(lambda: [0, 0.0])
with open(path) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name") or r.get("Name")
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg[name]
        a[0] += 1
        a[1] += d
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot/1e3:.2f} ms over {sum(v[0] for v in agg.values())} launches")
print(f"{'calls':>8} {'total_ms':>10} {'avg_us':>9} {'%':>6}  kernel")
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    short = name if len(name) < 150 else name[:147] + "..."
    print(f"{n:8d} {t/1e3:10.3f} {t/n:9.2f} {100*t/tot:6.2f}  {short}")
