#!/bin/bash
export NPP_BENCH_SUPERVISE=0
export GPU_MAX_HW_QUEUES=2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-prof > /tmp/tr.log 2>&1
tail -1 /tmp/tr.log | cut -c1-100
mkdir -p gpurun_out/timeline; python3 tools/step_timeline.py /tmp/tr gpurun_out/timeline 1.0 > gpurun_out/timeline/summary.txt 2>&1
head -12 gpurun_out/timeline/summary.txt
