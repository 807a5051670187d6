#!/bin/bash
# GPU box: the profiling artefacts judged for a round -> gpurun_out/final/ (copy into profiles/ afterwards)
#   1. rocprofv3 --kernel-trace --stats of the default bench command (hipGraph, two branch streams)
#   2. the same for the eager single-stream run (per-kernel durations without overlap)
#   3. two --pmc passes (FETCH_SIZE, WRITE_SIZE) of the eager single-stream run -> HBM bytes per conv_s1 / conv_g8 launch
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/final; mkdir -p $out
run() {  # name, extra env assignment string, bench args...
  name=$1; shift
  rm -rf /tmp/fp_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp_$name -- python3 bench.py "$@" > $out/${tag}_${name}.log 2>&1
  f=$(find /tmp/fp_$name -name "*kernel_trace.csv" | head -1)
  python3 tools/prof_summary.py "$f" > $out/${tag}_kernel_trace_${name}.txt 2>&1
  s=$(find /tmp/fp_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$s" ] && head -40 "$s" > $out/${tag}_kernel_stats_${name}.csv
  tail -1 $out/${tag}_${name}.log | cut -c1-300
}
run bs16_bf16_graph --steps 10 --warmup 3 --no-cpu-baseline
export NPP_STREAMS=1
run bs16_bf16_eager_1stream --steps 5 --warmup 2 --no-cpu-baseline --graph 0
# (NPP_SYNC_LAUNCH: see tools/pmc_step.sh)
export NPP_SYNC_LAUNCH=1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fp_pmc_$c
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/fp_pmc_$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --graph 0 > /dev/null 2>&1
done
ff=$(find /tmp/fp_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
fw=$(find /tmp/fp_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$ff" "$fw" conv_s1_kernel,conv_g8_kernel,conv_g4_kernel,conv_wgrad_g4_kernel $out/${tag}_pmc_traffic.json
python3 tools/pmc_step_total.py "$ff" "$fw" 4 > $out/${tag}_pmc_step_total.txt
