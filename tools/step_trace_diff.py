"""Two rocprofv3 kernel traces (CSV) -> the kernels of ONE step each (between two consecutive pack_weights_batched_kernel launches),
grouped by kernel base name: launches and total duration side by side, sorted by the difference."""
import collections
import csv
import re
import sys


def one_step(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "pack_weights_batched_kernel" in r["Kernel_Name"]]
    # (the 4th step of the run: bench.py --warmup 2 has timed steps there; the LAST steps of a --force-dist run are its ablations)
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    lo, hi = marks[k], marks[k + 1]
    fam = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[lo:hi]:
        nm = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        base = re.split(r"[<(]", nm)[0]
        if base.startswith("at::native::") or base == "":
            base = nm[:60]
        f = fam[base]
        f[0] += 1
        f[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    wall = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e6
    return fam, wall


a, wa = one_step(sys.argv[1])
b, wb = one_step(sys.argv[2])
print(f"A {sys.argv[1]}: {sum(v[0] for v in a.values())} launches, {sum(v[1] for v in a.values()) / 1e3:.2f} ms kernel time, step {wa:.2f} ms")
print(f"B {sys.argv[2]}: {sum(v[0] for v in b.values())} launches, {sum(v[1] for v in b.values()) / 1e3:.2f} ms kernel time, step {wb:.2f} ms")
names = sorted(set(a) | set(b), key=lambda n: -abs(a.get(n, [0, 0])[1] - b.get(n, [0, 0])[1]))
for n in names[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    x, y = a.get(n, [0, 0.0]), b.get(n, [0, 0.0])
    print(f"{n[:60]:60s} A {x[0]:5d} {x[1] / 1e3:7.3f} ms | B {y[0]:5d} {y[1] / 1e3:7.3f} ms | {(x[1] - y[1]) / 1e3:+7.3f}")
