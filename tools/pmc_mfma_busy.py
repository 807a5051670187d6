"""MFMA utilisation per kernel family from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
SQ_INSTS_MFMA ...): sums over every launch of the family in the run.
  busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES   (share of the cycles in which the shader engines had work at all during
                                                     which an MFMA pipe was executing; both are summed over SEs/XCDs)
usage: pmc_mfma_busy.py <counters.csv> <family>[,<family>...]      family = name=sub1|sub2 or a bare substring"""
import csv
import sys
from collections import defaultdict

path, fams = sys.argv[1], sys.argv[2].split(",")
rows = list(csv.DictReader(open(path)))
print(f"{'family':22s} {'launches':>8s} {'MFMA_BUSY':>14s} {'SQ_BUSY':>14s} {'busy':>7s}   other counters (sum)")
for fam in fams:
    name, _, pat = fam.rpartition("=")
    name = name or pat
    agg, n = defaultdict(float), defaultdict(int)
    for r in rows:
        if any(q in r["Kernel_Name"] for q in pat.split("|")):
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
    if not agg:
        print(f"{name:22s} (no launches)")
        continue
    mb, sb = agg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), agg.get("SQ_BUSY_CYCLES", 0.0)
    rest = "  ".join(f"{k}={v:.3g}" for k, v in sorted(agg.items()) if k not in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"))
    print(f"{name:22s} {max(n.values()):8d} {mb:14.4g} {sb:14.4g} {(mb / sb if sb else float('nan')):7.3f}   {rest}")
