#!/bin/bash
# GPU box: everything judged under profiles/r04_* in one call (copy gpurun_out/final/* and the files named below into profiles/)
cd $GRAFT_REPO_ROOT
bash tools/final_profiles.sh r04 > gpurun_out/final_r04.log 2>&1; echo "final rc=$?"; tail -3 gpurun_out/final_r04.log | cut -c1-300
GPU_MAX_HW_QUEUES=2 timeout -k 10 300 python3 tools/phase_stamps.py 2>&1 | grep -v amdgpu > gpurun_out/r04_phase_stamps.txt; echo "stamps rc=$?"
bash tools/g8_ablation.sh > /dev/null 2>&1; cp gpurun_out/g8_ablation.txt gpurun_out/r04_g8_ablation.txt; echo "g8 rc=$?"
bash tools/shape_prof.sh _r04 > /dev/null 2>&1; echo "shape rc=$?"; head -5 gpurun_out/shape_prof_r04.txt
