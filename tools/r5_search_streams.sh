#!/bin/bash
# GPU box: config 5 (supernet weights pass, batch 8): one / two branch streams x hardware queues of the hipGraph executor
cd $GRAFT_REPO_ROOT
one() { python3 bench.py --model search --batch 8 --no-cpu-baseline --no-prof --steps 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config'].get('hw_queues'))"; }
echo -n "streams 2, default queues:        "; one
echo -n "streams 1, default queues:        "; NPP_STREAMS=1 one
echo -n "streams 1, 1 queue (unsupervised): "; NPP_STREAMS=1 NPP_BENCH_SUPERVISE=0 GPU_MAX_HW_QUEUES=1 one
echo -n "streams 2, 3 queues (unsupervised): "; NPP_BENCH_SUPERVISE=0 GPU_MAX_HW_QUEUES=3 one
echo -n "streams 2, default queues:        "; one
