"""bench.py's per-rank supervisor (N > 1 runs: VERDICT r2 item 6a) without a GPU: a worker that stops making progress is
ended (by its own PID) and replaced by a fresh worker -- first on the collective SyncBatchNorm transport, then eager (`--graph 0`) -- whose record carries `fallback`; a worker lost AFTER it
has written its record does not cost the measurement; a healthy worker's line is relayed unchanged."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(fake, timeout=60):
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29731",
               NPP_BENCH_FAKE_WORKER=fake, NPP_BENCH_STALL_S="2", NPP_BENCH_STALL_FIRST_S="2")
    env.pop("NPP_BENCH_WORKER", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_healthy_worker_is_relayed():
    r, rec = _run("ok")
    assert r.returncode == 0 and rec["attempt"] == 0 and "fallback" not in rec


@pytest.mark.parametrize("fake", ["hang", "crash"])
def test_stalled_or_dead_worker_is_replaced_by_one_with_stand_alone_exchange_kernels(fake):
    r, rec = _run(fake)
    assert r.returncode == 0, r.stderr[-500:]
    assert rec["attempt"] == 1 and rec["p2p_fold"] == "0" and rec["syncbn_p2p"] == "1" and rec["graph_arg"] != 0 and "fallback" in rec
    assert "supervisor" in r.stderr


def test_a_second_stall_ends_on_the_collective_transport():
    r, rec = _run("hang2", timeout=90)
    assert r.returncode == 0, r.stderr[-500:]
    assert rec["attempt"] == 2 and rec["syncbn_p2p"] == "0" and rec["graph_arg"] != 0 and "collective" in rec["fallback"]


def test_a_third_stall_ends_in_the_eager_worker():
    r, rec = _run("hang3", timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    assert rec["attempt"] == 3 and rec["graph_arg"] == 0 and rec["syncbn_p2p"] == "0" and "eager" in rec["fallback"]


def test_worker_lost_after_its_record_keeps_the_record():
    r, rec = _run("late_hang")
    assert r.returncode == 0 and rec["attempt"] == 0


def _run_single(fake, timeout=60):
    env = dict(os.environ, NPP_BENCH_FAKE_WORKER=fake, NPP_BENCH_STALL_S="2", NPP_BENCH_STALL_FIRST_S="2")
    for k in ("NPP_BENCH_WORKER", "WORLD_SIZE", "RANK", "LOCAL_RANK", "GPU_MAX_HW_QUEUES"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_single_gpu_line_runs_on_two_hardware_queues_with_a_default_queue_worker_behind_it():
    """The default 1-GPU run: a child on GPU_MAX_HW_QUEUES=2; if that child dies (a capture on fewer hardware queues than captured
    streams has crashed the runtime for the supernet) a fresh child on the runtime's default delivers the line, marked `fallback`."""
    r, rec = _run_single("ok")
    assert r.returncode == 0 and rec["attempt"] == 0 and rec["hw_queues"] == "2" and "fallback" not in rec
    r, rec = _run_single("crash")
    assert r.returncode == 0, r.stderr[-500:]
    assert rec["attempt"] == 1 and rec["hw_queues"] is None and "default queues" in rec["fallback"]


def test_an_exported_queue_count_is_left_alone():
    env = dict(os.environ, NPP_BENCH_FAKE_WORKER="ok", GPU_MAX_HW_QUEUES="4")
    for k in ("NPP_BENCH_WORKER", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                       text=True, timeout=60)
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert r.returncode == 0 and rec["hw_queues"] == "4" and rec["attempt"] == 0 and "supervisor" not in r.stderr
