cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_syncbn_gpu.py tests/test_comm_gpu.py tests/test_p2p_gpu.py tests/test_configs_gpu.py tests/test_train_step_gpu.py -q --durations=8 > gpurun_out/t_sync.log 2>&1; echo "sync rc=$?"; tail -16 gpurun_out/t_sync.log
