#!/bin/bash
# GPU box: LDS counters of the weight-gradient kernels (one --pmc pass over tools/wgrad_time.py) -> gpurun_out/wgrad_lds_pmc.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/wgpmc
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/wgpmc -- python3 tools/wgrad_time.py > gpurun_out/wgrad_lds_pmc.log 2>&1
f=$(find /tmp/wgpmc -name "*counter_collection.csv" | head -1)
python3 - "$f" > gpurun_out/wgrad_lds_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
    if "wgrad" not in k: continue
    key = (k, r.get("Grid_Size", ""))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
for key, v in acc.items():
    d = {c: v[c] / max(n[(key, c)], 1) for c in v}
    conf, act = d.get("SQ_LDS_BANK_CONFLICT", 0.0), d.get("SQ_LDS_IDX_ACTIVE", 0.0)
    print(key, ", ".join(f"{c} {d[c]:.4g}" for c in sorted(d)), f"conflict/active {conf / act:.3f}" if act else "")
PY
cat gpurun_out/wgrad_lds_pmc.txt
