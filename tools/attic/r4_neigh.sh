#!/bin/bash
export NPP_BENCH_SUPERVISE=0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export NPP_STREAMS=1
rm -rf /tmp/ng; rocprofv3 --kernel-trace --output-format csv -d /tmp/ng -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --graph 0 > /tmp/ng.log 2>&1
tail -1 /tmp/ng.log | cut -c1-120
f=$(find /tmp/ng -name "*kernel_trace.csv" | head -1)
for pat in copyBuffer "FillFunctor<float>" "FillFunctor<c10::BFloat16>" channel_stats_kernel add_n_kernel sum_replicas CatArray; do
  echo "=== $pat"; python3 tools/neighbours.py "$f" "$pat" | head -14
done > gpurun_out/neigh.txt 2>&1
