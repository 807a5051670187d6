#!/bin/bash
export NPP_BENCH_SUPERVISE=0   # under rocprofv3 the profiled process must be the worker itself: never a supervisor that spawns one
# usage: tools/prof_run.sh <tag> <bench args...>   (GPU box) -> gpurun_out/<tag>_summary.txt + <tag>.log
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 bench.py "$@" > gpurun_out/$tag.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 tools/prof_summary.py "$f" > gpurun_out/${tag}_summary.txt 2>&1
s=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$s" ] && head -60 "$s" > gpurun_out/${tag}_kernel_stats_head.csv
tail -1 gpurun_out/$tag.log
