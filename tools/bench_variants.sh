#!/bin/bash
# GPU box: the bench lines of the other modes / configurations -> gpurun_out/<round>_bench_variants.txt (copy into profiles/)
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r02}_bench_variants.txt
: > $out
run() {
  echo "# python bench.py $*" >> $out
  timeout -k 10 400 python bench.py "$@" 2> gpurun_out/variant.err | tail -1 >> $out || echo "{\"failed\": \"$*\"}" >> $out
  tail -1 $out | cut -c1-200
}
C="--no-cpu-baseline --no-prof --steps 10"
run --force-dist $C
NPP_P2P_ALONE=1 run --force-dist $C
echo "# (the line above: NPP_P2P_ALONE=1 -- the peer-to-peer exchange kernels in the chain, 1-rank mailbox)" >> $out
NPP_P2P_ALONE=1 run --force-dist --batch 32 $C
echo "# (the line above: NPP_P2P_ALONE=1)" >> $out
run --launcher eager $C
run --launcher auto $C
run --batch 32 $C
run --batch 8 --size 512 $C
run --model search --batch 8 $C
run --model search --alpha-pass --batch 8 $C
run --model search --launcher auto --batch 8 $C
