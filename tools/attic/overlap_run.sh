#!/bin/bash
export NPP_BENCH_SUPERVISE=0   # under rocprofv3 the profiled process must be the worker itself: never a supervisor that spawns one
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/ov; rocprofv3 --kernel-trace --output-format csv -d /tmp/ov -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-prof > /tmp/ov.log 2>&1
f=$(find /tmp/ov -name "*kernel_trace.csv" | head -1)
python3 tools/overlap_stats.py "$f" 0.5 0.95
