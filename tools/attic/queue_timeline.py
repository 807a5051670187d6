"""Per-millisecond occupancy of the HIP queues over ONE replayed step, from a rocprofv3 --kernel-trace CSV of the bench run:
for every 1-ms bin of the last full step (pack_weights_batched_kernel to the next one) the fraction of the bin each queue has a
kernel in flight, and the fraction with two in flight.   usage: queue_timeline.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "pack_weights_batched_kernel" in r["Kernel_Name"]]
t0, t1 = marks[-2], marks[-1]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in rows
      if t0 <= int(r["Start_Timestamp"]) < t1]
queues = sorted({k[2] for k in ks}, key=lambda q: -sum(1 for k in ks if k[2] == q))
print(f"step {(t1 - t0) / 1e6:.2f} ms, {len(ks)} launches, queues {queues}")
nb = int((t1 - t0) / 1e6) + 1
busy = {q: [0.0] * nb for q in queues}
for s, e, q, _ in ks:
    b = int((s - t0) / 1e6)
    while s < e and b < nb:
        lim = t0 + (b + 1) * 1000000
        seg = min(e, lim) - s
        busy[q][b] += seg / 1e6
        s += seg
        b += 1
# two in flight: sweep
ev = []
for s, e, q, _ in ks:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
two = [0.0] * nb
depth, last = 0, t0
for t, d in ev:
    if depth >= 2:
        s = last
        while s < t:
            b = int((s - t0) / 1e6)
            lim = t0 + (b + 1) * 1000000
            seg = min(t, lim) - s
            if b < nb:
                two[b] += seg / 1e6
            s += seg
    last, depth = t, depth + d
print(" ms   " + "  ".join(f"q{q:>3}" for q in queues) + "   two   longest kernel starting in the bin")
for b in range(nb):
    inbin = [k for k in ks if int((k[0] - t0) / 1e6) == b]
    big = max(inbin, key=lambda k: k[1] - k[0]) if inbin else None
    name = (big[3].split("(")[0][-48:] + f" {(big[1] - big[0]) / 1e3:.0f}us") if big else ""
    print(f"{b:3d}   " + "  ".join(f"{busy[q][b]:4.2f}" for q in queues) + f"   {two[b]:4.2f}   {len(inbin):4d} launches  {name}")
tot = {q: sum(busy[q]) for q in queues}
print("kernel ms per queue:", {q: round(v, 2) for q, v in tot.items()}, " two in flight:", round(sum(two), 2))
