#!/bin/bash
# GPU box: one-box A/B of a few dispatch knobs on the current build -> gpurun_out/knob_sweep.txt (ms per step, 20 steps each)
cd $GRAFT_REPO_ROOT
out=gpurun_out/knob_sweep.txt; : > $out
run() { echo -n "$* : " >> $out; env "$@" python bench.py --no-cpu-baseline --no-prof --steps 20 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" >> $out; }
run NPP_X=0
run NPP_G4_PERS=2
run NPP_G4_PERS=0
run NPP_G4_BIG_MIN_TILES=512
run NPP_G4_BIG_MIN_TILES=128
run NPP_H3_MIN_PIX=9000
run NPP_G8_MIN_TILES=64
run NPP_WGB_MAX_BLOCKS=192
run NPP_WGB_MAX_BLOCKS=384
run NPP_C32_MIN_TILES=1000000
run NPP_X=0
cat $out
