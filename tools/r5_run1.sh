#!/bin/bash
# GPU box, round 5 first run: exactness of the weight-gradient kernels (asm LDS-DMA, nine-tap halo kernel), then timings.
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5_run1; mkdir -p $o
timeout -k 10 500 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "weight_gradient or batched_small or BENCH or conv_cases or lean" > $o/pytest.log 2>&1; echo "pytest rc $?" | tee -a $o/pytest.log
tail -5 $o/pytest.log
timeout -k 10 200 python3 tools/wgrad_time.py > $o/wgrad_time.txt 2>&1; cat $o/wgrad_time.txt
for w in 1 0; do NPP_WG9=$w timeout -k 10 200 python3 tools/wgrad_batched_time.py > $o/wgb_wg9_$w.txt 2>&1; tail -1 $o/wgb_wg9_$w.txt; done
for st in 32 128; do NPP_WG9_STAGES=$st timeout -k 10 200 python3 tools/wgrad_batched_time.py > $o/wgb_stages_$st.txt 2>&1; tail -1 $o/wgb_stages_$st.txt; done
NPP_WG9=1 timeout -k 10 200 python3 tools/wgrad_batched_time.py all > $o/wgb_all_1.txt 2>&1; tail -1 $o/wgb_all_1.txt
NPP_WG9=0 timeout -k 10 200 python3 tools/wgrad_batched_time.py all > $o/wgb_all_0.txt 2>&1; tail -1 $o/wgb_all_0.txt
timeout -k 10 300 python3 tools/g8_time.py 16 > $o/g8_time.txt 2>&1; cat $o/g8_time.txt
