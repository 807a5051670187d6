#!/bin/bash
# GPU box: A/B of the inline-asm LDS-DMA against the compiler's builtin (NPP_DMA_BUILTIN) and of conv_g8's per-tile wait (G8_EPI_WAIT),
# same box, rebuilt objects.  -> gpurun_out/r5_dma_ab.txt
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_dma_ab.txt; : > $out
run() {
  echo "== $1" >> $out
  touch npp_amd/csrc/conv_g8.hip npp_amd/csrc/conv_wgrad_g4.hip
  NPP_EXTRA_HIPCC_FLAGS="$2" bash npp_amd/csrc/build.sh > /dev/null 2>&1 || { echo "build failed: $1"; exit 1; }
  NPP_TIME_SET=ab timeout -k 10 200 python3 tools/g8_time.py 16 2>&1 | grep "k1" >> $out || exit 1
  timeout -k 10 200 python3 tools/g8_time_dgrad.py 16 2>&1 | grep "k1\|->" | head -6 >> $out
  timeout -k 10 200 python3 tools/wgrad_time.py 2>&1 | grep "k3\|k1" | head -8 >> $out || exit 1
}
run "asm DMA, epilogue wait (default)" ""
run "builtin DMA" "-DNPP_DMA_BUILTIN=1 -DG8_EPI_WAIT=0"
run "asm DMA, no epilogue wait" "-DG8_EPI_WAIT=0"
run "asm DMA, epilogue wait (default) again" ""
touch npp_amd/csrc/conv_g8.hip npp_amd/csrc/conv_wgrad_g4.hip
bash npp_amd/csrc/build.sh > /dev/null 2>&1
cat $out
