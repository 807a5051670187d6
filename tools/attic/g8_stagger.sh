#!/bin/bash
# GPU box: A/B of the conv_g8 start stagger / store drain -> gpurun_out/g8_stagger.txt
cd $GRAFT_REPO_ROOT
out=gpurun_out/g8_stagger.txt; : > $out
for cfg in "0 0 1" "3 0 1" "3 1000 1" "3 2000 1" "3 2000 0" "0 2000 1" "3 3000 1" "3 0 1"; do
  set -- $cfg
  echo "drain=$1 stagger_ns=$2 key=$3" >> $out
  NPP_G8_DRAIN=$1 NPP_G8_STAGGER_NS=$2 NPP_G8_STAGGER_KEY=$3 NPP_G8_DBG=0 timeout -k 10 200 python3 tools/g8_ablation.py 2>&1 | grep "^dbg" >> $out || exit 1
done
cat $out
