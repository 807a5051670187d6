"""Which kernels run just before / after a named kernel in a rocprofv3 --kernel-trace CSV (single-stream eager run)."""
import csv, sys
from collections import Counter
path, needle = sys.argv[1], sys.argv[2]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), (r.get("Kernel_Name") or r.get("Name"))))
rows.sort()
short = lambda s: s.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
c = Counter()
for i, (_, n) in enumerate(rows):
    if needle in n:
        prev = short(rows[i - 1][1]) if i else "-"
        nxt = short(rows[i + 1][1]) if i + 1 < len(rows) else "-"
        c[(prev, nxt)] += 1
for (p, n), k in c.most_common(25):
    print(f"{k:6d}  {p}  ->  [{needle}]  ->  {n}")
