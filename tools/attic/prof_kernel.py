"""Per-launch durations of the kernels whose name contains argv[2], grouped by grid size (rocprofv3 kernel-trace CSV)."""
import csv, sys
from collections import defaultdict
path, pat = sys.argv[1], sys.argv[2]
agg = defaultdict(list)
with open(path) as f:
    for r in csv.DictReader(f):
        if pat in r["Kernel_Name"]:
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            agg[(r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("LDS_Block_Size"), r.get("VGPR_Count"))].append(d)
print(f"{'grid':>10} {'lds':>7} {'vgpr':>5} {'n':>5} {'avg_us':>9} {'min':>8} {'max':>8} {'total_ms':>9}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:>10} {k[1]:>7} {k[2]:>5} {len(v):5d} {sum(v)/len(v):9.2f} {min(v):8.2f} {max(v):8.2f} {sum(v)/1e3:9.3f}")
