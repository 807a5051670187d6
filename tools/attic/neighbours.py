"""Which kernels surround the launches whose name contains <pattern> (same queue, by start time) in a rocprofv3 kernel trace:
finds the call sites of stray ATen fills / copies in the eager step.  usage: neighbours.py <kernel_trace.csv> <pattern>"""
import csv
import sys
from collections import Counter

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2]
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
names = [short(r["Kernel_Name"]) for r in rows]
c = Counter()
sizes = Counter()
for i, r in enumerate(rows):
    if pat in r["Kernel_Name"]:
        prev = names[i - 1] if i else "-"
        nxt = names[i + 1] if i + 1 < len(rows) else "-"
        c[(prev, nxt)] += 1
        sizes[(r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"))] += 1
print("launches:", sum(c.values()))
for (p, n), k in c.most_common(25):
    print(f"{k:6d}  after {p:60s}  before {n}")
print("grid sizes:", sizes.most_common(8))
