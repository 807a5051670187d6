cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_wide_gpu.py -x -q > gpurun_out/t_wide.log 2>&1; echo "wide rc=$?"; tail -4 gpurun_out/t_wide.log
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py tests/test_train_step_gpu.py -x -q > gpurun_out/t_ops.log 2>&1; echo "ops rc=$?"; tail -4 gpurun_out/t_ops.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/b3.json 2> gpurun_out/b3.err; echo "bench rc=$?"; tail -c 300 gpurun_out/b3.json
NPP_SE_PAIR=0 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/b3_nose.json 2> gpurun_out/b3_nose.err; echo "bench nose rc=$?"; tail -c 200 gpurun_out/b3_nose.json
