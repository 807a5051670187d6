#!/bin/bash
# GPU box: ablation timings of conv_g8_kernel<256> on the 1024->512 1x1 head shape (NPP_G8_DBG bits: 1 no epilogue, 2 no MFMA, 4 no DMA, 8 no ds_read)
for d in 0 1 2 3 4 5 7 9 13 15; do
  NPP_G8_DBG=$d NPP_G8_ONLY=1 timeout -k 10 120 python tools/g8_time.py 16 one 2>&1 | grep -v amdgpu.ids | sed "s/^/dbg=$d /"
done
