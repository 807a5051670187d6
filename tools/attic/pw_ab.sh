#!/bin/bash
# GPU box: A/B of a conv_g4.hip compile-time switch on the memory-bound 1x1 shapes (tools/pw_time.py): tools/pw_ab.sh "-DG4_RING=2" "-DG4_RING=3"
for flags in "$@"; do
  touch npp_amd/csrc/conv_g4.hip
  NPP_EXTRA_HIPCC_FLAGS="$flags" bash npp_amd/csrc/build.sh > /dev/null 2>&1
  timeout -k 10 200 python tools/pw_time.py 2>&1 | grep "TB/s" | sed "s/^/[$flags] /"
done
