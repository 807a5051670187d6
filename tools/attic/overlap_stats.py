"""Concurrency profile of a rocprofv3 kernel trace: share of wall time with 0 / 1 / 2+ kernels in flight, over the
last `nsteps` repetitions found (uses the largest gaps as step separators is overkill: just takes a time window)."""
import csv, sys
path = sys.argv[1]
frac0, frac1 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5, float(sys.argv[3]) if len(sys.argv) > 3 else 0.9
ev = []
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
w0, w1 = t0 + (t1 - t0) * frac0, t0 + (t1 - t0) * frac1
for s, e, n in rows:
    if e < w0 or s > w1:
        continue
    ev.append((max(s, w0), 1)); ev.append((min(e, w1), -1))
ev.sort()
depth, last = 0, w0
hist = {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    last = t
    depth += d
hist[depth] = hist.get(depth, 0) + (w1 - last)
tot = w1 - w0
print(f"window {tot/1e6:.1f} ms")
for k in sorted(hist):
    print(f"  {k} kernels in flight: {100*hist[k]/tot:5.1f} %")
