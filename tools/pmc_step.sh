#!/bin/bash
export NPP_BENCH_SUPERVISE=0   # under rocprofv3 the profiled process must be the worker itself: never a supervisor that spawns one
# GPU box: whole-step HBM traffic (two --pmc passes of the eager single-stream bench) -> gpurun_out/pmc_step_total${1}.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; export NPP_STREAMS=1
# NPP_SYNC_LAUNCH: with a deep launch queue rocprofv3's counter-collection intercept aborts the queue ("AQL packet is
# malformed", seen once the eager host got faster); waiting for every launch keeps the queue shallow, per-kernel counters are unaffected
export NPP_SYNC_LAUNCH=1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pm_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --graph 0 > /dev/null 2>&1
done
ff=$(find /tmp/pm_FETCH_SIZE -name "*counter_collection.csv" | head -1); fw=$(find /tmp/pm_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_step_total.py $ff $fw 3 > gpurun_out/pmc_step_total${1}.txt
head -${2:-24} gpurun_out/pmc_step_total${1}.txt
