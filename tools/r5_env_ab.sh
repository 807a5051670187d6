#!/bin/bash
# GPU box: the default bench step with environment settings, each measured twice, interleaved:  tools/r5_env_ab.sh "A=1" "A=2 B=3" ...
cd $GRAFT_REPO_ROOT
one() { python3 bench.py --no-cpu-baseline --no-prof --steps ${STEPS:-20} 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
  echo -n "base: "; one
  for setting in "$@"; do
    echo -n "$setting: "; env $setting bash -c "$(declare -f one); one"
  done
done
