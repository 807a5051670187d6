#!/bin/bash
# GPU box: A/B of conv_g4's software-pipelined K loop (G4_PIPE) on the starved-grid shapes, same box, rebuilt object.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_g4_pipe_ab.txt; : > $out
run() {
  echo "== G4_PIPE=$1" >> $out
  touch npp_amd/csrc/conv_g4.hip
  NPP_EXTRA_HIPCC_FLAGS="-DG4_PIPE=$1" bash npp_amd/csrc/build.sh > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  NPP_TIME_SET=g4b timeout -k 10 200 python3 tools/g8_time.py 16 2>&1 | grep "relu+stats=1" >> $out || exit 1
  NPP_TIME_SET=g4 timeout -k 10 200 python3 tools/g8_time.py 16 2>&1 | grep "relu+stats=1" >> $out || exit 1
  NPP_TIME_SET=g4b timeout -k 10 200 python3 tools/g8_time_dgrad.py 2>&1 | grep "dgrad" >> $out
}
run 1; run 0; run 1
touch npp_amd/csrc/conv_g4.hip; bash npp_amd/csrc/build.sh > /dev/null 2>&1
cat $out
