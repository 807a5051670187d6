#!/bin/bash
# GPU box: A/B of a conv_g8.hip compile-time switch on ONE device: tools/g8_ab.sh "-DG8_SPLIT=0" "-DG8_SPLIT=1"
for flags in "$@"; do
  touch npp_amd/csrc/conv_g8.hip npp_amd/csrc/conv_g4.hip
  NPP_EXTRA_HIPCC_FLAGS="$flags" bash npp_amd/csrc/build.sh > /dev/null 2>&1
  for r in 1 2; do
    NPP_TIME_SET=${NPP_TIME_SET:-ab} timeout -k 10 200 python tools/g8_time.py 2>&1 | grep "k1\|k3" | sed "s/^/[$flags] /"
  done
done
