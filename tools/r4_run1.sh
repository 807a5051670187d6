cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1 0 1; do
NPP_TAIL_SPLIT=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/b5_$v.json 2> gpurun_out/b5.err; echo "split=$v rc=$?"; python3 -c "
import json;d=json.loads(open('gpurun_out/b5_$v.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"
done
