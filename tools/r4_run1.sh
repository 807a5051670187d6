cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_wide_gpu.py -x -q > gpurun_out/t_wide.log 2>&1; echo "wide rc=$?"; tail -4 gpurun_out/t_wide.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/b2.json 2> gpurun_out/b2.err; echo "bench rc=$?"; tail -c 300 gpurun_out/b2.json
NPP_BN_MULTI=0 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/b2_nomulti.json 2> gpurun_out/b2_nomulti.err; echo "bench nomulti rc=$?"; tail -c 300 gpurun_out/b2_nomulti.json
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1; echo "all rc=$?"; tail -8 gpurun_out/t_all.log
