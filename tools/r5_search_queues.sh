#!/bin/bash
# GPU box: config 5 (supernet weights pass) on 1 / 2 / 4 / 8 hardware queues
cd $GRAFT_REPO_ROOT
for q in 2 4 8 1; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  NPP_BENCH_SUPERVISE=0 GPU_MAX_HW_QUEUES=$q python3 bench.py --model search --batch 8 --no-cpu-baseline --no-prof --steps 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
