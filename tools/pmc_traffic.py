"""Average per-launch HBM traffic of kernel families from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-byte requests as 64 bytes for wide
(16 B/lane) streaming reads -> doubled; WRITE_SIZE is exact.  Units: KB in the CSV.
A family is `name=sub1|sub2|...` (every launch whose kernel name contains one of the substrings) or a bare substring.
The JSON records the hash of the kernel sources the counters were collected on (npp_amd._lib.kernel_source_hash): bench.py
refuses the numbers once the kernels change.
usage: pmc_traffic.py <fetch.csv> <write.csv> <family>[,<family>...] <out.json>"""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npp_amd._lib import kernel_source_hash      # noqa: E402
fetch_csv, write_csv, pats, out = sys.argv[1:5]


def avg(path, counter, pat):
    vals = []
    for r in csv.DictReader(open(path)):
        if any(q in r["Kernel_Name"] for q in pat.split("|")) and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (0.0, 0)


db = {"source_hash": kernel_source_hash(), "collected_by": "tools/final_profiles.sh (two rocprofv3 --pmc passes of "
      "`bench.py --graph 0`, NPP_STREAMS=1)"}
for fam in pats.split(","):
    name, _, pat = fam.rpartition("=")
    name = name or pat
    f, nf = avg(fetch_csv, "FETCH_SIZE", pat)
    w, nw = avg(write_csv, "WRITE_SIZE", pat)
    db[name] = {"kernel": pat, "launches": nf, "fetch_kb_raw": f, "write_kb": w,
               "traffic_bytes_per_launch": (2.0 * f + w) * 1024.0,
               "note": "FETCH_SIZE doubled (gfx950 wide-read under-count), WRITE_SIZE exact; separate --pmc passes of `bench.py --graph 0`"}
json.dump(db, open(out, "w"), indent=1)
print(json.dumps(db))
