#!/bin/bash
# GPU box: per-shape conv times of the training step -> gpurun_out/shape_prof.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/sp; rocprofv3 --kernel-trace --output-format csv -d /tmp/sp -- python3 tools/shape_prof.py run /tmp/sp_log.json > /tmp/sp_run.log 2>&1 || tail -20 /tmp/sp_run.log
f=$(find /tmp/sp -name "*kernel_trace.csv" | head -1)
python3 tools/shape_prof.py join /tmp/sp_log.json "$f" > gpurun_out/shape_prof${1}.txt 2>&1
head -70 gpurun_out/shape_prof${1}.txt
