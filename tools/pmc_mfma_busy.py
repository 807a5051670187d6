"""MFMA utilisation per kernel family from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
SQ_INSTS_MFMA ...): sums over every launch of the family in the run.
  util = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES)
SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 MFMA pipes (256 CUs x 4 SIMDs; = 16 x SQ_INSTS_MFMA for
v_mfma_f32_16x16x32_bf16, MI355X_MICROARCH.md "s_memtime tick vs SQ PMC units"), SQ_BUSY_CYCLES the busy cycles of the 32 shader
engines (8 XCDs x 4): per launch it is 32 x the kernel's duration in shader clocks (checked against the kernel trace), so
1024 / 32 = 32 pipes per counted SQ cycle.  util is the share of the kernel's own duration, at the clock it actually ran at,
in which an MFMA pipe was executing -- the counter-side twin of bench.py's roofline.frac (which prices against 2.4 GHz peak).
usage: pmc_mfma_busy.py <counters.csv> <family>[,<family>...]      family = name=sub1|sub2 or a bare substring"""
import csv
import sys
from collections import defaultdict

path, fams = sys.argv[1], sys.argv[2].split(",")
rows = list(csv.DictReader(open(path)))
print(f"{'family':22s} {'launches':>8s} {'MFMA_BUSY':>14s} {'SQ_BUSY':>14s} {'util':>7s}   other counters (sum)")
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
    print(f"{name:22s} {max(n.values()):8d} {mb:14.4g} {sb:14.4g} {(mb / (32 * sb) if sb else float('nan')):7.3f}   {rest}")
