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


import bg_children      # noqa: E402  (the probes are child processes: registered here, run in the background, see tests/bg_children.py)

_SO = os.path.join(REPO, "tools", "libpoison_alloc.so")


def _reg_probe(key, args, env_extra):
    return bg_children.register(key, [sys.executable, os.path.join(REPO, "tools", "poison_probe.py")] + args,
                                dict({"PROBE_STEPS": "4"}, **env_extra), timeout=600)


def _build_allocator():      # (before the first child starts: every probe loads it)
    if not os.path.isfile(_SO):
        subprocess.check_call(["hipcc", "-shared", "-fPIC", "-o", _SO, os.path.join(REPO, "tools", "poison_alloc.cpp")])


bg_children.before_start(_build_allocator)


def _probe(key, nsteps):
    r = bg_children.result(key)
    assert r.returncode == 0 and "PROBE_DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    steps = [ln for ln in r.stdout.splitlines() if ln.startswith("step")]
    assert len(steps) == int(nsteps)
    for ln in steps:
        assert "non-finite grads 0 [] params 0 [] buffers 0 []" in ln and "nan" not in ln.split("non-finite")[0], ln


_STEP_CASES = [(d, f) for d in ["f32", "bf16"] for f in ["1", "0"]]
for _d, _f in _STEP_CASES:
    _reg_probe(f"poison-step-{_d}-{_f}", ["bf16"] if _d == "bf16" else [], {"NPP_FANOUT": _f})


@pytest.mark.parametrize("dtype,fanout", _STEP_CASES)
def test_training_step_reads_no_unwritten_memory(dtype, fanout):
    _probe(f"poison-step-{dtype}-{fanout}", 4)


_OTHER_CASES = [("search", "f32", "3"), ("search", "bf16", "3"), ("syncbn", "f32", "3"), ("syncbn", "bf16", "3"), ("full", "bf16", "2")]
for _m, _d, _n in _OTHER_CASES:
    _reg_probe(f"poison-{_m}-{_d}", ["bf16"] if _d == "bf16" else [], {"PROBE_MODE": _m, "PROBE_STEPS": _n})


@pytest.mark.parametrize("mode,dtype,steps", _OTHER_CASES)
def test_other_paths_read_no_unwritten_memory(mode, dtype, steps):
    """search: SearchStep (weights pass + alpha pass with the entropy term) on the supernet; syncbn: SyncBatchNorm + GradReducer on a
    1-rank RCCL group; full: the C=64 network at 384 x 384 (the kernels the bench runs: conv_g8, conv_h3, conv_wgrad_g4 ...)."""
    _probe(f"poison-{mode}-{dtype}", steps)


_GRAPH_CASES = [("1", "0", "f32"), ("1", "1", "f32"), ("0", "0", "f32"), ("1", "0", "bf16")]
for _s, _o, _d in _GRAPH_CASES:
    bg_children.register(f"graph-poison-{_s}-{_o}-{_d}",
                         [sys.executable, os.path.join(REPO, "tools", "graph_poison_probe.py")] + (["bf16"] if _d == "bf16" else []),
                         dict(PROBE_SYNC=_s, PROBE_OVERLAP=_o, PROBE_GB="2"), timeout=600)
bg_children.register("graph-poison-p2p", [sys.executable, os.path.join(REPO, "tools", "graph_poison_probe.py"), "bf16"],
                     dict(PROBE_SYNC="1", PROBE_OVERLAP="0", PROBE_GB="2", NPP_P2P_ALONE="1", NPP_P2P_SELFTEST="200"), timeout=600)
bg_children.register("graph-poison-search", [sys.executable, os.path.join(REPO, "tools", "graph_poison_probe.py")],
                     dict(PROBE_MODEL="search", PROBE_GB="2"), timeout=600)


@pytest.mark.parametrize("sync,overlap,dtype", _GRAPH_CASES)
def test_replayed_step_reads_no_recycled_block(sync, overlap, dtype):
    """tools/graph_poison_probe.py: the hipGraph-replayed step (SyncBatchNorm + GradReducer on a 1-rank RCCL group: hub streams,
    lockstep issue; or local BatchNorm on two streams) leaves no non-finite value and no wild gradient element behind -- checked on
    gradients, parameters, buffers AND the optimizer state (a wild finite element only shows as exp_avg_sq = inf).  Regression for
    the cross-stream use-after-free of round 2 (an input read on the hub stream without record_stream)."""
    r = bg_children.result(f"graph-poison-{sync}-{overlap}-{dtype}")
    assert r.returncode == 0 and "GRAPH_PROBE_DONE" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_replayed_step_with_the_exchanges_inside_the_fused_kernels():
    """The same probe with the SyncBatchNorm statistics going through the (1-rank) peer-to-peer mailboxes: the exchanges run inside the
    fused BatchNorm kernels' prologues (csrc/p2p_xp.h), captured and replayed as part of the step's hipGraph -- sequence numbers,
    result vectors and done counters advance on the device from replay to replay.  No non-finite value, no mailbox error, and the
    exchanges really were folded."""
    r = bg_children.result("graph-poison-p2p")
    assert r.returncode == 0 and "GRAPH_PROBE_DONE" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    line = next(ln for ln in r.stdout.splitlines() if ln.startswith("folded exchanges"))
    fwd, bwd = (int(v) for v in line.split("[")[1].split("]")[0].split(","))
    assert fwd > 50 and bwd > 50 and "p2p ok active True" in line, line


def test_replayed_search_step_reads_no_recycled_block():
    """The same probe on the supernet under SearchStep (both passes replayed as hipGraphs, two branch streams)."""
    r = bg_children.result("graph-poison-search")
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
