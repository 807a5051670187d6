#!/bin/bash
# GPU box: 1-rank rehearsal of the N > 1 path by the workgroup budget of a launch that carries an exchange (NPP_XP_MAX_BLOCKS)
cd $GRAFT_REPO_ROOT
C="--no-cpu-baseline --no-prof --steps 10 --force-dist"
one() { python3 bench.py $C "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
export NPP_P2P_ALONE=1
for rep in 1 2; do
  for b in 512 256 192 128; do echo -n "budget $b b16: "; NPP_XP_MAX_BLOCKS=$b one; done
done
for b in 512 192; do echo -n "budget $b b32: "; NPP_XP_MAX_BLOCKS=$b one --batch 32; done
