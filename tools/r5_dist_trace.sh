#!/bin/bash
# GPU box: eager single-stream kernel traces of the 1-rank rehearsal (A) and of the plain 1-GPU step (B), one step of each compared
export NPP_BENCH_SUPERVISE=0 GPU_MAX_HW_QUEUES=2 NPP_STREAMS=1 NPP_P2P_SELFTEST=0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/tda /tmp/tdb
NPP_P2P_ALONE=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/tda -- python3 bench.py --force-dist --steps 4 --warmup 2 --graph 0 --no-cpu-baseline --no-prof > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tdb -- python3 bench.py --steps 4 --warmup 2 --graph 0 --no-cpu-baseline --no-prof > /dev/null 2>&1
python3 tools/step_trace_diff.py $(find /tmp/tda -name "*kernel_trace.csv" | head -1) $(find /tmp/tdb -name "*kernel_trace.csv" | head -1) 45
