"""C-ABI collectives (include/npp_hip.h npp_comm_* / npp_allreduce_bucket / npp_syncbn_exchange) on one GPU: a world of one
rank is a real RCCL communicator -- the calls, dtypes, stream ordering and capture behaviour are the ones N ranks use; only the
arithmetic is trivial.  The N>1 arithmetic (mean over ranks, global statistics) is covered with gloo in test_ddp_cpu.py /
test_syncbn_gpu.py and by bench.py --gpus N on the 8-GPU node."""
import ctypes

import pytest
import torch

from npp_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture()
def comm1():
    lib = _lib.lib()
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    buf = ctypes.create_string_buffer(128)
    _lib.check(lib.npp_comm_unique_id(buf), "id")
    assert any(buf.raw), "unique id is empty"
    _lib.check(lib.npp_comm_init(bytes(buf.raw), 0, 1), "init")
    assert lib.npp_comm_world() == 1
    yield lib
    torch.cuda.synchronize()
    _lib.check(lib.npp_comm_destroy(), "destroy")
    assert lib.npp_comm_world() == 0


def test_bucket_and_statistics_exchange_world_of_one(comm1):
    lib = comm1
    st = torch.cuda.Stream()
    g = torch.randn(1 << 20, device="cuda")
    ref = g.clone()
    d = torch.randn(4096, device="cuda", dtype=torch.float64)
    dref = d.clone()
    h = torch.randn(1 << 16, device="cuda").bfloat16()
    href = h.clone()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g.mul_(2.0)       # ordering: the collective must see this
        _lib.check(lib.npp_allreduce_bucket(g.data_ptr(), g.numel(), _lib.NPP_F32, 1, st.cuda_stream), "ar")
        _lib.check(lib.npp_syncbn_exchange(d.data_ptr(), d.numel(), st.cuda_stream), "sx")
        _lib.check(lib.npp_allreduce_bucket(h.data_ptr(), h.numel(), _lib.NPP_BF16, 0, st.cuda_stream), "ar16")
    st.synchronize()
    assert torch.equal(g, ref * 2.0) and torch.equal(d, dref) and torch.equal(h, href)


def test_errors_without_communicator_and_double_init():
    lib = _lib.lib()
    torch.zeros(1, device="cuda")
    x = torch.zeros(8, device="cuda")
    assert lib.npp_comm_world() == 0
    assert lib.npp_allreduce_bucket(x.data_ptr(), 8, _lib.NPP_F32, 1, None) == _lib.NPP_E_RCCL
    assert b"npp_comm_init" in lib.npp_last_error()
    buf = ctypes.create_string_buffer(128)
    _lib.check(lib.npp_comm_unique_id(buf), "id")
    assert lib.npp_comm_init(bytes(buf.raw), 3, 2) == _lib.NPP_E_SHAPE
    _lib.check(lib.npp_comm_init(bytes(buf.raw), 0, 1), "init")
    try:
        assert lib.npp_comm_init(bytes(buf.raw), 0, 1) == _lib.NPP_E_UNSUPPORTED
        assert lib.npp_allreduce_bucket(x.data_ptr(), 8, 7, 1, None) == _lib.NPP_E_DTYPE
    finally:
        _lib.check(lib.npp_comm_destroy(), "destroy")


def test_collective_inside_a_hipgraph(comm1):
    lib = comm1
    g = torch.ones(1 << 18, device="cuda")
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        _lib.check(lib.npp_allreduce_bucket(g.data_ptr(), g.numel(), _lib.NPP_F32, 1, st.cuda_stream), "warm")
    st.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=st):
        g.add_(1.0)
        _lib.check(lib.npp_allreduce_bucket(g.data_ptr(), g.numel(), _lib.NPP_F32, 1, torch.cuda.current_stream().cuda_stream), "cap")
        g.mul_(2.0)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    # capture does not execute: 1 -> (1+1)*2 = 4 -> 10 -> 22 over three replays
    assert torch.equal(g, torch.full_like(g, 22.0)), g[:4]


import os
import bg_children      # noqa: E402  (child processes, run in the background: tests/bg_children.py)
import sys as _sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)
bg_children.register("comm-reducer", [_sys.executable, os.path.join(_HERE, "comm_reducer_worker.py")], timeout=600)
bg_children.register("reducer-defer", [_sys.executable, os.path.join(_REPO, "tools", "reducer_defer_check.py")], timeout=600)
bg_children.register("reducer-tail", [_sys.executable, os.path.join(_REPO, "tools", "reducer_defer_check.py")],
                     dict(NPP_CHECK_REDUCER_MODE="tail"), timeout=600)


def test_training_step_on_the_library_transport():
    """GradReducer + SyncBatchNorm + hipGraph on npp_allreduce_bucket / npp_syncbn_exchange (tests/comm_reducer_worker.py)."""
    r = bg_children.result("comm-reducer")
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_deferred_gradients_land_in_the_reducer_buckets():
    """overlap=False reducer + TrainStep's deferred / batched weight gradients (tools/reducer_defer_check.py): every gradient is a
    view of its bucket and equals a plain backward's."""
    r = bg_children.result("reducer-defer")
    assert r.returncode == 0 and "REDUCER_DEFER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_tail_overlap_reducer_two_group_tail():
    """overlap="tail": buckets per kind; after the KxK group of the batched weight-gradient tail (+ unpack) the KxK buckets are
    reduced, the 1x1 / depthwise / SE group follows, finish() reduces the rest -- every gradient equals a plain backward's."""
    r = bg_children.result("reducer-tail")
    assert r.returncode == 0 and "REDUCER_DEFER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
