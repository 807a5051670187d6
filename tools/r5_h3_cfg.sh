#!/bin/bash
# GPU box: conv_h3's tile configurations on the 3x3 shapes of the step (NPP_H3_CFG: 0 = 8 rows 4 waves 2 blocks/CU (default; 6 rows on
# small grids), 1 = 16 rows 8 waves, 3 = 12 rows 8 waves, 4 = 8 rows 8 waves, 6 = 6 rows 4 waves), forward, N = 16
cd $GRAFT_REPO_ROOT
for c in 0 3 1 4 6 0 3; do
  echo "== NPP_H3_CFG=$c"
  NPP_H3_CFG=$c NPP_TIME_SET=h3 timeout -k 10 200 python3 tools/g8_time.py 16 2>&1 | grep "TF/s" || exit 1
done
