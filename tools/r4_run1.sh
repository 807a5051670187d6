cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_train_step_gpu.py -x -q -k "dwconv or train or batched" > gpurun_out/t_ts.log 2>&1; echo "ts rc=$?"; tail -3 gpurun_out/t_ts.log
for v in 1 2; do
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/b11_$v.json 2> gpurun_out/b11.err; echo "rc=$?"; python3 -c "
import json;d=json.loads(open('gpurun_out/b11_$v.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['config']['loss'])"
done
