#!/bin/bash
# GPU box: split-K of conv_g4's starved grids (NPP_G4_SPLITK: 0 never, 2 / 3 / 4 forced, unset = the launch heuristic), forward, N = 16
cd $GRAFT_REPO_ROOT
for sk in 0 auto 2 3 4 0 auto; do
  echo "== NPP_G4_SPLITK=$sk"
  if [ $sk = auto ]; then NPP_TIME_SET=g4b timeout -k 10 200 python3 tools/g8_time.py 16 2>&1 | grep "TF/s" || exit 1
  else NPP_G4_SPLITK=$sk NPP_TIME_SET=g4b timeout -k 10 200 python3 tools/g8_time.py 16 2>&1 | grep "TF/s" || exit 1; fi
done
