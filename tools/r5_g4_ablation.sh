#!/bin/bash
# GPU box: where a starved-grid conv_g4 launch spends its time (G4_DBG: 1 no stores, 2 no operand DMA, 4 no MFMA), rebuilt objects, one box.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_g4_ablation.txt; : > $out
for d in 0 2 4 6 7; do
  echo "== G4_DBG=$d" >> $out
  touch npp_amd/csrc/conv_g4.hip
  NPP_EXTRA_HIPCC_FLAGS="-DG4_DBG=$d" bash npp_amd/csrc/build.sh > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  NPP_TIME_SET=g4b timeout -k 10 200 python3 tools/g8_time.py 16 2>&1 | grep "relu+stats=0" | grep "24^2\|12^2" >> $out || exit 1
done
touch npp_amd/csrc/conv_g4.hip; bash npp_amd/csrc/build.sh > /dev/null 2>&1
cat $out
