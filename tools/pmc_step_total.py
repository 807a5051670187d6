"""HBM bytes of a whole training step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; eager single-stream run):
total per step and the kernels that move the most.  Same corrections as tools/pmc_traffic.py (FETCH_SIZE doubled on gfx950).
usage: pmc_step_total.py <fetch.csv> <write.csv> <steps in the run>"""
import csv, sys
from collections import defaultdict
fetch_csv, write_csv, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
tot = defaultdict(lambda: [0.0, 0.0, 0])
for path, counter, idx in ((fetch_csv, "FETCH_SIZE", 0), (write_csv, "WRITE_SIZE", 1)):
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:90]
            tot[k][idx] += float(r["Counter_Value"]) * 1024.0 * (2.0 if idx == 0 else 1.0)
            if idx == 0:
                tot[k][2] += 1
rd = sum(v[0] for v in tot.values()) / steps
wr = sum(v[1] for v in tot.values()) / steps
print(f"HBM bytes per step: read {rd / 1e9:.2f} GB  write {wr / 1e9:.2f} GB  total {(rd + wr) / 1e9:.2f} GB  (over {steps} steps incl. warm-up)")
for k, v in sorted(tot.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:30]:
    print(f"{(v[0] + v[1]) / steps / 1e9:7.2f} GB  r {v[0] / steps / 1e9:6.2f}  w {v[1] / steps / 1e9:6.2f}  n/step {v[2] // steps:5d}  {k}")
