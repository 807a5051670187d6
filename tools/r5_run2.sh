#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5_run2; mkdir -p $o
timeout -k 10 500 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "weight_gradient or batched_small" > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log
tail -4 $o/pytest.log
bash tools/r5_dma_ab.sh
