cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 4 8 4 8 3; do
NPP_G4_DEEP_RING=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/b14_$v.json 2> gpurun_out/b14.err; echo "ring=$v rc=$?"; python3 -c "
import json;d=json.loads(open('gpurun_out/b14_$v.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"
done
