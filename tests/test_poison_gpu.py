"""Uninitialised / out-of-bounds reads.  tools/poison_probe.py runs the tiny training step on an allocator that gives every
tensor its own hipMalloc filled with 0xFF bytes (NaN as f32 / bf16): a kernel that consumes memory nobody wrote -- the row
padding of a tensor a producer left unwritten, bytes past the end of a buffer -- turns the loss, a gradient or a parameter into
NaN.  (Round 2: the data gradient of FactorizedReduce's second conv read 16 bytes past the end of the concatenation's gradient;
NaN x 0 = NaN, and a fault when the buffer closed a mapped segment.)"""
import os
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

pytestmark = pytest.mark.gpu


def _probe(args, env_extra):
    so = os.path.join(REPO, "tools", "libpoison_alloc.so")
    if not os.path.isfile(so):
        subprocess.check_call(["hipcc", "-shared", "-fPIC", "-o", so, os.path.join(REPO, "tools", "poison_alloc.cpp")])
    env = dict(os.environ, **{"PROBE_STEPS": "4", **env_extra})
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "poison_probe.py")] + args, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "PROBE_DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    steps = [ln for ln in r.stdout.splitlines() if ln.startswith("step")]
    assert len(steps) == int(env["PROBE_STEPS"])
    for ln in steps:
        assert "non-finite grads 0 [] params 0 [] buffers 0 []" in ln and "nan" not in ln.split("non-finite")[0], ln


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("fanout", ["1", "0"])
def test_training_step_reads_no_unwritten_memory(dtype, fanout):
    _probe(["bf16"] if dtype == "bf16" else [], {"NPP_FANOUT": fanout})


@pytest.mark.parametrize("mode,dtype,steps", [("search", "f32", "3"), ("search", "bf16", "3"), ("syncbn", "f32", "3"),
                                              ("syncbn", "bf16", "3"), ("full", "bf16", "2")])
def test_other_paths_read_no_unwritten_memory(mode, dtype, steps):
    """search: SearchStep (weights pass + alpha pass with the entropy term) on the supernet; syncbn: SyncBatchNorm + GradReducer on a
    1-rank RCCL group; full: the C=64 network at 384 x 384 (the kernels the bench runs: conv_g8, conv_h3, conv_wgrad_g4 ...)."""
    _probe(["bf16"] if dtype == "bf16" else [], {"PROBE_MODE": mode, "PROBE_STEPS": steps})


@pytest.mark.parametrize("sync,overlap,dtype", [("1", "0", "f32"), ("1", "1", "f32"), ("0", "0", "f32"), ("1", "0", "bf16")])
def test_replayed_step_reads_no_recycled_block(sync, overlap, dtype):
    """tools/graph_poison_probe.py: the hipGraph-replayed step (SyncBatchNorm + GradReducer on a 1-rank RCCL group: hub streams,
    lockstep issue; or local BatchNorm on two streams) leaves no non-finite value and no wild gradient element behind -- checked on
    gradients, parameters, buffers AND the optimizer state (a wild finite element only shows as exp_avg_sq = inf).  Regression for
    the cross-stream use-after-free of round 2 (an input read on the hub stream without record_stream)."""
    env = dict(os.environ, PROBE_SYNC=sync, PROBE_OVERLAP=overlap, PROBE_GB="2", MASTER_PORT="29637")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "graph_poison_probe.py")] + (["bf16"] if dtype == "bf16" else []),
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "GRAPH_PROBE_DONE" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_replayed_search_step_reads_no_recycled_block():
    """The same probe on the supernet under SearchStep (both passes replayed as hipGraphs, two branch streams)."""
    env = dict(os.environ, PROBE_MODEL="search", PROBE_GB="2")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "graph_poison_probe.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "GRAPH_PROBE_DONE" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_channel_slice_at_the_end_of_a_buffer_is_repacked():
    """conv inputs are read in groups of 8 channels: the upper half [4:8] of an 8-wide buffer would be read past its end."""
    from npp_amd import _ops as K
    buf = torch.randn(2, 6, 6, 8, device="cuda").permute(0, 3, 1, 2)        # logical NCHW over NHWC memory
    lo, hi = buf[:, 0:4], buf[:, 4:8]
    assert K._gemm_ready(lo) is lo                      # the group read of the last pixel ends exactly at the buffer's end
    ready = K._gemm_ready(hi)
    assert ready is not hi and torch.equal(ready, hi)
    from npp_amd import _lib as L
    assert L.nhwc_ld(ready) >= 8
    whole = torch.randn(2, 6, 6, 16, device="cuda").permute(0, 3, 1, 2)
    mid = whole[:, 4:8]                                 # a slice with room behind it stays in place
    assert K._gemm_ready(mid) is mid
