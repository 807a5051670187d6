#!/bin/bash
# GPU box: conv_g8 with its operand half-tiles sourced out of range (NPP_G8_DBG 16: weights, 32: x, 48: both; +1: no epilogue) --
# the same DMA instructions, LDS writes, reads and MFMAs, but nothing fetched: what the L2 / HBM side of each operand stream costs
cd $GRAFT_REPO_ROOT
for d in 0 16 32 48 1 17 33 4; do
  NPP_G8_DBG=$d timeout -k 10 200 python3 tools/g8_ablation.py 2>&1 | grep "^dbg" | grep -v full || exit 1
done
