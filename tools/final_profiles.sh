#!/bin/bash
export NPP_BENCH_SUPERVISE=0   # under rocprofv3 the profiled process must be the worker itself: never a supervisor that spawns one
# GPU box: the profiling artefacts judged for a round -> gpurun_out/final/ (copy into profiles/ afterwards)
#   1. rocprofv3 --kernel-trace --stats of the default bench command (hipGraph, two branch streams)
#   2. the same for the eager single-stream run (per-kernel durations without overlap)
#   3. two --pmc passes (FETCH_SIZE, WRITE_SIZE) of the eager single-stream run -> HBM bytes per conv_s1 / conv_g8 launch
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# (the profiled program must be the worker itself -- no supervisor child under rocprofv3 -- on the queue count the bench line uses)
export GPU_MAX_HW_QUEUES=2
out=gpurun_out/final; mkdir -p $out
run() {  # name, extra env assignment string, bench args...
  name=$1; shift
  rm -rf /tmp/fp_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp_$name -- python3 bench.py "$@" > $out/${tag}_${name}.log 2>&1
  f=$(find /tmp/fp_$name -name "*kernel_trace.csv" | head -1)
  python3 tools/prof_summary.py "$f" > $out/${tag}_kernel_trace_${name}.txt 2>&1
  s=$(find /tmp/fp_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$s" ] && cp "$s" $out/${tag}_kernel_stats_${name}.csv      # every row
  tail -1 $out/${tag}_${name}.log | cut -c1-300
}
run bs16_bf16_graph --steps 10 --warmup 3 --no-cpu-baseline
python3 tools/family_graph_ms.py $out/${tag}_kernel_stats_bs16_bf16_graph.csv $out/${tag}_family_graph_ms.json
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
FAMS="conv_g4=conv_g4_kernel|conv_h3_kernel|conv_thin_out_kernel|conv_thin_in_kernel|conv_c32_kernel,conv_g8=conv_g8_kernel,conv_wgrad=conv_wgrad_g4_kernel|conv_wgrad_g4_batched_kernel|conv_wgrad_g9_batched_kernel|conv_wgrad_narrow|conv_wgrad_kernel|conv_wgrad_s1_kernel|conv_wgrad_h3_kernel|conv_wgrad_thin_kernel,conv_h3_kernel,conv_g4_kernel,conv_wgrad_g4_batched_kernel,conv_wgrad_g9_batched_kernel"
python3 tools/pmc_traffic.py "$ff" "$fw" "$FAMS" $out/${tag}_pmc_traffic.json
python3 tools/pmc_step_total.py "$ff" "$fw" 4 > $out/${tag}_pmc_step_total.txt
#   4. MFMA utilisation per conv family (north_star: "rocprof HBM GB/s and MFMA utilisation")
rm -rf /tmp/fp_pmc_mfma
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d /tmp/fp_pmc_mfma -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --graph 0 > /dev/null 2>&1
fm=$(find /tmp/fp_pmc_mfma -name "*counter_collection.csv" | head -1)
python3 tools/pmc_mfma_busy.py "$fm" "$FAMS,bn_bwd,add_n,affine_add,dw" > $out/${tag}_pmc_mfma_busy.txt
cat $out/${tag}_pmc_mfma_busy.txt
#   5. the plain bench line (no profiler, default streams), after copying the fresh traffic file where bench.py reads it
unset NPP_STREAMS NPP_SYNC_LAUNCH GPU_MAX_HW_QUEUES NPP_BENCH_SUPERVISE
cp $out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json
cp $out/${tag}_family_graph_ms.json profiles/${tag}_family_graph_ms.json
python3 bench.py > $out/${tag}_bench_line.json 2> /dev/null; tail -c 400 $out/${tag}_bench_line.json
