#!/bin/bash
# GPU box: config 5 (supernet weights pass, batch 8) with an environment switch off / on, A/B/A/B:  tools/r5_search_ab.sh VAR=VALUE
cd $GRAFT_REPO_ROOT
one() { python3 bench.py --model search --batch 8 --no-cpu-baseline --no-prof --steps 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
  echo -n "base:    "; one
  echo -n "$1: "; env "$@" bash -c "$(declare -f one); one"
done
