"""One replayed training step out of a rocprofv3 kernel trace: concurrency histogram, time per kernel family split by how many
kernels were in flight, per-queue busy time, and a coarse timeline (what ran in every `bucket` ms).  Writes the step's rows to
<out>/step_rows.csv for offline analysis.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-prof
    python3 tools/step_timeline.py /tmp/tr gpurun_out/timeline [bucket_ms]
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
bucket = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
path = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_]+)(<[^(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:60]


adam = [i for i, r in enumerate(rows) if "adam_multi" in r[2].lower()]
assert len(adam) >= 3, "need at least three optimizer launches in the trace"
lo, hi = adam[-2] + 1, adam[-1] + 1          # the last complete step: after the previous Adam, through this step's Adam
step = rows[lo:hi]
t0, t1 = step[0][0], max(r[1] for r in step)
os.makedirs(out, exist_ok=True)
with open(os.path.join(out, "step_rows.csv"), "w") as f:
    f.write("start_us,end_us,queue,kernel\n")
    for s, e, n, q in step:
        f.write(f"{(s - t0) / 1e3:.2f},{(e - t0) / 1e3:.2f},{q},{short(n)}\n")
lines = [f"step: {len(step)} kernels, {(t1 - t0) / 1e6:.2f} ms wall, {sum(e - s for s, e, _, _ in step) / 1e6:.2f} ms of kernel time"]
# concurrency
ev = []
for s, e, n, q in step:
    ev.append((s, 1, n))
    ev.append((e, -1, n))
ev.sort(key=lambda x: (x[0], x[1]))
depth, last = 0, t0
hist = defaultdict(float)
for t, d, n in ev:
    hist[depth] += t - last
    last = t
    depth += d
for k in sorted(hist):
    lines.append(f"  {k} kernels in flight: {hist[k] / 1e6:7.2f} ms  ({100 * hist[k] / (t1 - t0):5.1f} %)")
# per queue
perq = defaultdict(float)
for s, e, n, q in step:
    perq[q] += e - s
for q, v in sorted(perq.items()):
    lines.append(f"  queue {q}: busy {v / 1e6:7.2f} ms")
# time by family, alone vs overlapped
fam_alone, fam_tot, fam_n = defaultdict(float), defaultdict(float), defaultdict(int)
active = {}
last = t0
ev2 = []
for i, (s, e, n, q) in enumerate(step):
    ev2.append((s, 1, i))
    ev2.append((e, -1, i))
ev2.sort(key=lambda x: (x[0], x[1]))
for t, d, i in ev2:
    if len(active) == 1:
        k = next(iter(active))
        fam_alone[short(step[k][2])] += t - last
    last = t
    if d == 1:
        active[i] = True
    else:
        active.pop(i, None)
for s, e, n, q in step:
    fam_tot[short(n)] += e - s
    fam_n[short(n)] += 1
lines.append("  family: launches, total ms, ms running ALONE")
for k, v in sorted(fam_tot.items(), key=lambda kv: -kv[1])[:45]:
    lines.append(f"    {k[:70]:70s} {fam_n[k]:5d} {v / 1e6:7.2f} {fam_alone[k] / 1e6:7.2f}")
# coarse timeline
nb = int((t1 - t0) / 1e6 / bucket) + 1
busy = [defaultdict(float) for _ in range(nb)]
for s, e, n, q in step:
    a, b = (s - t0) / 1e6, (e - t0) / 1e6
    i = int(a / bucket)
    while i < nb and i * bucket < b:
        ov = min(b, (i + 1) * bucket) - max(a, i * bucket)
        if ov > 0:
            busy[i][short(n).split("<")[0]] += ov
        i += 1
lines.append(f"  timeline ({bucket} ms buckets): kernel-ms in bucket | top families")
for i, d in enumerate(busy):
    tot = sum(d.values())
    top = ", ".join(f"{k} {v:.2f}" for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:4])
    lines.append(f"    {i * bucket:6.1f} ms: {tot:5.2f} | {top}")
txt = "\n".join(lines)
print(txt)
open(os.path.join(out, "summary.txt"), "w").write(txt + "\n")
