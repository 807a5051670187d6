"""Kernel time per conv family and step inside the REPLAYED graph run, from rocprofv3's kernel_stats.csv (tools/final_profiles.sh):
    python3 tools/family_graph_ms.py kernel_stats.csv out.json
Steps covered by the trace = calls of pack_weights_batched_kernel (once per step)."""
import csv
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npp_amd import _lib

FAMS = {"conv_g4": ("conv_g4_kernel", "conv_h3_kernel", "conv_thin_out_kernel", "conv_thin_in_kernel", "conv_c32_kernel"),
        "conv_g8": ("conv_g8_kernel",),
        "conv_wgrad": ("conv_wgrad_g4_kernel", "conv_wgrad_g4_batched_kernel", "conv_wgrad_g9_batched_kernel", "conv_wgrad_narrow", "conv_wgrad_kernel",
                       "conv_wgrad_s1_kernel", "conv_wgrad_h3_kernel", "conv_wgrad_thin_kernel", "wgrad_tap_kernel")}
rows = list(csv.DictReader(open(sys.argv[1])))
steps = next((int(r["Calls"]) for r in rows if "pack_weights_batched_kernel" in r["Name"]), 0)
out = {"source_hash": _lib.kernel_source_hash(), "steps_in_trace": steps, "file": os.path.basename(sys.argv[1])}
for fam, names in FAMS.items():
    ns = sum(float(r["TotalDurationNs"]) for r in rows if any(n in r["Name"] for n in names))
    calls = sum(int(r["Calls"]) for r in rows if any(n in r["Name"] for n in names))
    out[fam] = {"ms_per_step": round(ns / steps / 1e6, 3) if steps else None, "launches_per_step": round(calls / steps, 1) if steps else None}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
