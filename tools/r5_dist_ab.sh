#!/bin/bash
# GPU box: 1-rank rehearsal of the N > 1 path (--force-dist, NPP_P2P_ALONE=1): SyncBatchNorm exchanges as launches of their own
# (NPP_P2P_FOLD=0; also without the multi-job backward) against exchanges inside the fused kernels' prologues, A/B/A/B
cd $GRAFT_REPO_ROOT
C="--no-cpu-baseline --no-prof --steps 10 --force-dist"
one() { python3 bench.py $C "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
export NPP_P2P_ALONE=1
for rep in 1 2; do
  echo -n "fold off       b16: "; NPP_P2P_FOLD=0 one
  [ "$1" = "full" ] && { echo -n "multi-sync off b16: "; NPP_P2P_FOLD=0 NPP_BN_MULTI_SYNC=0 one; }
  echo -n "fold on        b16: "; one
done
echo -n "fold off       b32: "; NPP_P2P_FOLD=0 one --batch 32
echo -n "fold on        b32: "; one --batch 32
