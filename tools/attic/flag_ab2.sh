#!/bin/bash
# GPU box: A/B of compile flags on given sources: tools/flag_ab2.sh "src1.hip src2.hip" "<flags A>" "<flags B>" ... -> default bench x3 each
srcs=$1; shift
for flags in "$@"; do
  for s in $srcs; do touch npp_amd/csrc/$s; done
  NPP_EXTRA_HIPCC_FLAGS="$flags" bash npp_amd/csrc/build.sh > /dev/null 2>&1
  for r in 1 2 3; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-prof 2>/dev/null | tail -1 | cut -c80-130 | sed "s/^/[$flags] /"
  done
done
