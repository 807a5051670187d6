#!/bin/bash
# GPU box: A/B of a compile flag over ALL MFMA kernels: rebuild the conv sources with each flag set, then pw_time + the default bench
for flags in "$@"; do
  touch npp_amd/csrc/conv_g4.hip npp_amd/csrc/conv_g8.hip npp_amd/csrc/conv_s1.hip npp_amd/csrc/conv_wgrad_g4.hip npp_amd/csrc/conv_wgrad_s1.hip npp_amd/csrc/conv_igemm.hip npp_amd/csrc/conv_wgrad.hip
  NPP_EXTRA_HIPCC_FLAGS="$flags" bash npp_amd/csrc/build.sh > /dev/null 2>&1
  timeout -k 10 200 python tools/pw_time.py 2>&1 | grep "TB/s" | sed "s/^/[$flags] /"
  for r in 1 2; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-prof 2>/dev/null | tail -1 | cut -c80-160 | sed "s/^/[$flags] /"
  done
done
