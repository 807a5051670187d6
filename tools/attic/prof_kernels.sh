#!/bin/bash
# GPU box: kernel-trace of shape_prof's eager steps, then per-grid breakdown of the kernels named in "$@"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/sp; rocprofv3 --kernel-trace --output-format csv -d /tmp/sp -- python3 tools/shape_prof.py run /tmp/sp_log.json > /tmp/sp_run.log 2>&1 || tail -20 /tmp/sp_run.log
f=$(find /tmp/sp -name "*kernel_trace.csv" | head -1)
for k in "$@"; do echo "== $k"; python3 tools/prof_kernel.py "$f" "$k" | head -${NROWS:-14}; done
