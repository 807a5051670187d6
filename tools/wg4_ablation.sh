#!/bin/bash
# GPU box: ablation of the 128 x 128 weight-gradient kernel (conv_wgrad_g4.hip, WG4_DBG bits: 1 no epilogue, 2 no MFMA, 4 no DMA,
# 8 no fragment reads) -> gpurun_out/wg4_ablation.txt.  Rebuilds the one object per variant ON the box; restores the real one at the end.
cd $GRAFT_REPO_ROOT
out=gpurun_out/wg4_ablation.txt; : > $out
for d in 0 1 2 4 8 15; do
  touch npp_amd/csrc/conv_wgrad_g4.hip
  NPP_EXTRA_HIPCC_FLAGS="-DWG4_DBG=$d" bash npp_amd/csrc/build.sh > /dev/null 2>&1 || { echo "build failed for $d"; exit 1; }
  echo "WG4_DBG=$d" >> $out
  timeout -k 10 200 python3 tools/wgrad_time.py 2>&1 | grep "k3\|k1" | grep -v "384->   6\|32->  32\|64->  64" >> $out || exit 1
done
touch npp_amd/csrc/conv_wgrad_g4.hip
bash npp_amd/csrc/build.sh > /dev/null 2>&1
cat $out
