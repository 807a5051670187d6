#!/bin/bash
export NPP_BENCH_SUPERVISE=0   # under rocprofv3 the profiled process must be the worker itself: never a supervisor that spawns one
# usage: tools/pmc.sh "<kernel substring>" <python script + args...>   (run on the GPU box)
pat=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES"; do
  rm -rf /tmp/pm; rocprofv3 --pmc $c --output-format csv -d /tmp/pm -- python3 "$@" > /dev/null 2>&1
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$pat" <<PY
import csv,sys
from collections import defaultdict
agg=defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
done
