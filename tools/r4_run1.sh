cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q > gpurun_out/t_ts.log 2>&1; echo "ts rc=$?"; tail -5 gpurun_out/t_ts.log
for v in 1 0; do
NPP_RESAMPLE_SWAP=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/b10_$v.json 2> gpurun_out/b10.err; echo "swap=$v rc=$?"; python3 -c "
import json;d=json.loads(open('gpurun_out/b10_$v.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['config']['loss'])"
done
