"""Host side of the C-ABI collectives (include/npp_hip.h: npp_comm_init / npp_allreduce_bucket / npp_syncbn_exchange).

The default transport of GradReducer and of the SyncBatchNorm exchange is torch.distributed (backend "nccl" = RCCL); this module
is the other one: the library's own RCCL communicator, enqueued directly on the stream the caller is on -- no ProcessGroup work
objects, no internal stream, nothing but the collective in a captured graph.  It is what a non-torch host of libnpp_hip.so
would use, and `NPP_COMM=npp` (or `comm.enable()`) switches the training path to it.

torch.distributed is still the side channel for the 128-byte unique id (replaces ProcessGroupNCCL's store exchange behind
augment_lip_sync.py:68 init_process_group).

Second transport, for the SyncBatchNorm statistics only (csrc/p2p.hip, `enable_p2p`): a one-shot peer-to-peer exchange through
hipIpc-mapped mailboxes -- one small kernel per exchange instead of an all-reduce; on by default for N > 1 ranks of one node
(`NPP_SYNCBN_P2P=0` keeps the collectives), every rank falls back to the collective together if any rank cannot set it up.
"""
import os

import torch
import torch.distributed as dist

from . import _lib

_state = {"world": 0, "group": None}


def active():
    return _state["world"] > 0


def enable(group=None):
    """Collective: every rank of `group` (default: WORLD) joins one RCCL communicator owned by libnpp_hip.so on its current
    CUDA device.  Idempotent per process."""
    if active():
        return
    if not torch.cuda.is_available():
        raise RuntimeError("npp_amd.comm: the library's RCCL transport needs a GPU")
    lib = _lib.lib()
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    blob = [None]
    if rank == 0:
        import ctypes
        buf = ctypes.create_string_buffer(128)
        _lib.check(lib.npp_comm_unique_id(buf), "npp_comm_unique_id")
        blob[0] = bytes(buf.raw)
    if world > 1:
        dist.broadcast_object_list(blob, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    torch.cuda.current_stream().synchronize()
    _lib.check(lib.npp_comm_init(blob[0], rank, world), "npp_comm_init")
    _state["world"], _state["group"] = world, group


def disable():
    if active():
        torch.cuda.synchronize()
        _lib.check(_lib.lib().npp_comm_destroy(), "npp_comm_destroy")
        _state["world"], _state["group"] = 0, None


def wanted():
    return os.environ.get("NPP_COMM", "torch") == "npp"


def all_reduce_bucket(flat, average=True):
    """In-place all-reduce of one flat gradient bucket on the current stream."""
    assert flat.is_cuda and flat.is_contiguous() and flat.dtype in (torch.float32, torch.bfloat16)
    dt = _lib.NPP_F32 if flat.dtype == torch.float32 else _lib.NPP_BF16
    _lib.check(_lib.lib().npp_allreduce_bucket(flat.data_ptr(), flat.numel(), dt, 1 if average else 0,
                                               torch.cuda.current_stream().cuda_stream), "npp_allreduce_bucket")


def syncbn_exchange(stats):
    """In-place SUM of f64 BatchNorm partial sums on the current stream."""
    assert stats.is_cuda and stats.is_contiguous() and stats.dtype == torch.float64
    _lib.check(_lib.lib().npp_syncbn_exchange(stats.data_ptr(), stats.numel(), torch.cuda.current_stream().cuda_stream),
               "npp_syncbn_exchange")


# ---- one-shot peer-to-peer exchange of the SyncBatchNorm statistics (csrc/p2p.hip) ----------------------------------------------
_p2p = {"world": 0, "group": None, "cap": 0, "channels": {}, "nchan": 0, "count": 0, "mode": None}
P2P_CAP_DOUBLES = int(os.environ.get("NPP_P2P_CAP", str(1 << 15)))      # 256 KiB per (slot, source rank): a merged exchange of a whole
                                                                         # lockstep stage of model_augment at C = 64 is < 100 KiB


def p2p_active():
    return _p2p["world"] > 0


def p2p_wanted():
    return os.environ.get("NPP_SYNCBN_P2P", "1") != "0"


def enable_p2p(group=None, channels=4):
    """Collective over `group`: allocate the mailboxes, exchange the IPC handles through torch.distributed, map the peers.
    Returns True if EVERY rank succeeded (the ranks agree through a MIN all-reduce; otherwise all of them close their mailboxes
    and keep the collective transport).  Ranks of different hosts: not attempted."""
    if p2p_active():
        return True
    if not (torch.cuda.is_available() and dist.is_available() and dist.is_initialized()):
        return False
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world > 16 or (world < 2 and os.environ.get("NPP_P2P_ALONE") != "1"):      # (a 1-rank group: rehearsal of the kernel chain only)
        return False
    import ctypes
    import socket
    lib = _lib.lib()
    nb = int(lib.npp_p2p_handle_bytes())
    buf = ctypes.create_string_buffer(nb)
    ok = 1
    try:
        _lib.check(lib.npp_p2p_alloc(rank, world, P2P_CAP_DOUBLES, channels, buf), "npp_p2p_alloc")
    except Exception:      # noqa: BLE001  (no IPC on this runtime: every rank learns it below)
        ok = 0
    mine = (socket.gethostname(), bytes(buf.raw) if ok else b"")
    everyone = [None] * world
    dist.all_gather_object(everyone, mine, group=group)
    if ok and (any(h != mine[0] for h, _ in everyone) or any(len(b) != nb for _, b in everyone)):
        ok = 0
    if ok:
        try:
            _lib.check(lib.npp_p2p_open(b"".join(b for _, b in everyone)), "npp_p2p_open")
        except Exception:      # noqa: BLE001
            ok = 0
    backend = dist.get_backend(group)

    def agree(v):      # MIN over the ranks of one int
        flag = torch.tensor([v], dtype=torch.int32)
        if backend == "nccl":
            flag = flag.cuda()
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return int(flag.item())

    # everybody has mapped everybody before anybody exchanges
    if agree(ok) != 1:
        lib.npp_p2p_close()
        return False
    # acceptance test (2000 back-to-back exchanges, see _p2p_selftest).  The unit modes to try follow from what the allocator handed
    # out: uncached / fine-grained mailboxes (kinds 0, 1) try relaxed units first and release / acquire units if ANY rank read a
    # wrong sum or timed out; plain device memory (kind 2) is only ever run with release / acquire units -- a relaxed system-scope
    # poll of such memory may never observe the peer's stores, and a pass there would be luck, not coherence.  The test runs with a
    # short poll timeout (NPP_P2P_SELFTEST_TIMEOUT_MS, 5 s; the long watchdog value is restored afterwards), and between modes every
    # rank clears its channels' error words: a timed-out channel keeps counting its exchanges, so the ranks are still in step.
    mode = None
    forced = os.environ.get("NPP_P2P_LIGHT")
    kind = agree(int(lib.npp_p2p_alloc_kind()))      # (MIN over the ranks; mixed kinds: below the most careful rank's list is used)
    kind_max = -agree(-int(lib.npp_p2p_alloc_kind()))
    if forced is not None:
        order = (1 if forced != "0" else 0,)
    elif kind_max >= 2 or kind < 0:
        order = (0,)
    else:
        order = (1, 0)
    long_ms = int(lib.npp_p2p_set_timeout_ms(int(os.environ.get("NPP_P2P_SELFTEST_TIMEOUT_MS", "5000"))))
    try:
        for k, light in enumerate(order):
            if k > 0:
                if agree(1 if int(lib.npp_p2p_reset_errors()) == 0 else 0) != 1:
                    break
            lib.npp_p2p_set_mode(light)
            _selftest_state["broken"] = False
            passed = agree(1 if _p2p_selftest(lib, rank, world) else 0) == 1
            if passed:
                mode = "relaxed" if light else "fenced"
                break
            if agree(0 if _selftest_state["broken"] else 1) != 1:      # (a rank that did not issue every exchange: out of step)
                break
    finally:
        lib.npp_p2p_set_timeout_ms(long_ms)
    if mode is None:
        lib.npp_p2p_set_mode(-1)
        lib.npp_p2p_close()
        return False
    _p2p["mode"] = mode
    _p2p.update(world=world, group=group, cap=int(lib.npp_p2p_capacity()), channels={torch.cuda.current_stream().cuda_stream: 0},
                nchan=int(lib.npp_p2p_channels()), count=0)      # (channel 0 = the stream the self-test ran on)
    return True


SELFTEST_EXCHANGES = int(os.environ.get("NPP_P2P_SELFTEST", "2000"))


def _p2p_selftest(lib, rank, world, rounds=None) -> bool:
    """Acceptance test of the mailboxes, run right after they are mapped and before any statistic depends on them: `rounds`
    (default 2000) exchanges BACK TO BACK on channel 0 with no host synchronisation in between -- alternating the plain form, the slab
    form and the in-kernel form of the fused BatchNorm kernels (npp_p2p_exchange_folded_test), vectors that change every round and differ per rank, lengths that change (so every slot is reused hundreds of times with
    different contents) -- and only then are all sums compared with the exact expected values (integers below 2^53: f64 sums are
    exact in any order).  A stale read of an earlier exchange's slot, a torn unit, a missing peer or a mapping to the wrong memory
    fails here.  The caller's MIN all-reduce makes every rank drop the transport if one of them fails."""
    try:
        rounds = SELFTEST_EXCHANGES if rounds is None else rounds
        st = torch.cuda.current_stream().cuda_stream
        n, R = 1031, 16
        dev = torch.device("cuda", torch.cuda.current_device())
        idx = torch.arange(n, dtype=torch.float64, device=dev)
        # round `it`, rank r contributes  idx * (it + 1) + (r + 1) * (it + 3); the world's sum is known in closed form
        its = torch.arange(rounds, dtype=torch.float64, device=dev).view(-1, 1)
        mine = idx.view(1, -1) * (its + 1) + float(rank + 1) * (its + 3)                      # [rounds][n]
        want = idx.view(1, -1) * ((its + 1) * world) + (its + 3) * float(world * (world + 1) // 2)
        lens = [n - (it % 7) * 64 for it in range(rounds)]                                    # changing lengths
        plain = mine.clone()
        for it in range(rounds):                  # (elements past this round's length are not exchanged: preset to the expectation)
            plain[it, lens[it]:] = want[it, lens[it]:]
        outf = torch.zeros(n, dtype=torch.float32, device=dev)
        for it in range(rounds):
            ln = lens[it]
            if it % 4 == 2:
                # the in-kernel form (what the fused BatchNorm kernels carry in their prologues: leader workgroup, tagged result vector)
                _lib.check(lib.npp_p2p_exchange_folded_test(plain[it].data_ptr(), ln, 0, st), "npp_p2p_exchange_folded_test")
            elif it % 2 == 0:
                _lib.check(lib.npp_p2p_exchange(plain[it].data_ptr(), ln, 0, st), "npp_p2p_exchange")
            else:
                # slab form: the contribution split over R replica slabs [R][ln] (every slab floor(value / R), slab 0 the rest)
                base = torch.floor(mine[it, :ln] / R)
                seg = base.view(1, ln).repeat(R, 1).contiguous()
                seg[0] = mine[it, :ln] - base * (R - 1)
                arr = (_lib.NppP2pSeg * 1)()
                arr[0] = _lib.NppP2pSeg(seg.data_ptr(), ln, 0, outf.data_ptr(), None, None, None, R, 1, None)
                _lib.check(lib.npp_p2p_exchange_slabs(arr, 1, 0, st), "npp_p2p_exchange_slabs")
                plain[it, :ln] = seg.sum(0)       # (enqueued AFTER the exchange on the same stream: replica 0 = world sum, rest zero)
        if not torch.equal(plain, want):          # the first host synchronisation since the first exchange
            return False
        return int(lib.npp_p2p_status()) == 0
    except Exception:      # noqa: BLE001  (this rank may not have issued every exchange: the channel's sequence is out of step)
        _selftest_state["broken"] = True
        return False


_selftest_state = {"broken": False}
_p2p_tried: set = set()


def ensure_p2p(group=None) -> bool:
    """p2p_active() after ONE attempt per group to set the mailboxes up (a collective: call it where every rank is at the same
    point of the program -- the first SyncBatchNorm forward; never inside a hipGraph capture)."""
    if p2p_active():
        return group is _p2p["group"]
    if not p2p_wanted() or id(group) in _p2p_tried or not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
        return False
    _p2p_tried.add(id(group))
    return enable_p2p(group)


def disable_p2p():
    if _p2p["world"]:
        torch.cuda.synchronize()
        _lib.lib().npp_p2p_close()
        _p2p.update(world=0, group=None, cap=0, channels={}, nchan=0)


def _p2p_channel(create=True):
    """The mailbox channel of the current stream (one per stream, in order of first use -- the same order on every rank)."""
    st = torch.cuda.current_stream().cuda_stream
    ch = _p2p["channels"].get(st)
    if ch is None and create and len(_p2p["channels"]) < _p2p["nchan"]:
        ch = _p2p["channels"][st] = len(_p2p["channels"])
    return ch, st


def p2p_can(n_doubles, group=None) -> bool:
    """Would an exchange of n_doubles on the CURRENT stream go through the mailboxes?"""
    return p2p_active() and group is _p2p["group"] and n_doubles <= _p2p["cap"] and _p2p_channel()[0] is not None


def p2p_exchange(stats, group=None):
    """In-place SUM over the ranks on the current stream; False if this tensor / stream is not the mailboxes' (the caller then
    uses the collective).  Every stream gets a channel of its own in order of first use -- the same order on every rank."""
    if not p2p_active() or group is not _p2p["group"]:
        return False
    if not (stats.is_cuda and stats.is_contiguous() and stats.dtype == torch.float64):
        return False
    ch, st = _p2p_channel()
    if ch is None:
        return False
    cap, n = _p2p["cap"], stats.numel()
    for lo in range(0, n, cap):      # (a vector longer than a mailbox slot goes in pieces)
        _lib.check(_lib.lib().npp_p2p_exchange(stats.data_ptr() + 8 * lo, min(cap, n - lo), ch, st), "npp_p2p_exchange")
        _p2p["count"] += 1
    return True


def p2p_exchange_slabs(segs, group=None):
    """segs: up to 8 tuples (slabs f64 [nrep * len], len, nrep, split, (out0, out0_dup, out1, out2) f32 tensors | None, zero_rest)
    -- see npp_p2p_exchange_slabs.  False if the mailboxes cannot take it (the caller then reduces and exchanges separately)."""
    if not p2p_active() or group is not _p2p["group"] or not 1 <= len(segs) <= 8 or sum(s[1] for s in segs) > _p2p["cap"]:
        return False
    ch, st = _p2p_channel()
    if ch is None:
        return False
    arr = (_lib.NppP2pSeg * len(segs))()
    for k, (slabs, ln, nrep, split, outs, zero_rest) in enumerate(segs):
        assert slabs.is_cuda and slabs.dtype == torch.float64 and slabs.is_contiguous() and slabs.numel() >= nrep * ln
        o = [t.data_ptr() if t is not None else None for t in outs]
        arr[k] = _lib.NppP2pSeg(slabs.data_ptr(), ln, split, o[0], o[1], o[2], o[3], nrep, 1 if zero_rest else 0,
                                o[4] if len(o) > 4 else None)      # (outs[4]: every local sum as floats)
    _lib.check(_lib.lib().npp_p2p_exchange_slabs(arr, len(segs), ch, st), "npp_p2p_exchange_slabs")
    _p2p["count"] += 1
    return True


def p2p_fold_channel(n_doubles, group=None) -> int:
    """The channel on which a fused BatchNorm kernel of the CURRENT stream carries an exchange of n_doubles inside its prologue
    (npp_*_x entry points, csrc/p2p_xp.h), counted as an exchange; -1 if the mailboxes cannot take it."""
    if not p2p_active() or group is not _p2p["group"] or not 0 < n_doubles <= _p2p["cap"]:
        return -1
    ch, _st = _p2p_channel()
    if ch is None:
        return -1
    _p2p["count"] += 1
    return ch


def p2p_status() -> int:
    """0, or the error bits of the mailbox channels (1: a poll timed out, 2: a slot was overwritten before it was read).
    Synchronises the device -- every stream, the non-blocking branch / side streams the exchanges run on included."""
    if not p2p_active():
        return 0
    torch.cuda.synchronize()
    return int(_lib.lib().npp_p2p_status())


def p2p_ok():
    """False if a peer never showed up for some exchange (synchronises the device)."""
    return p2p_status() == 0


CHECK_EVERY = int(os.environ.get("NPP_P2P_CHECK_EVERY", "50"))      # steps between two health checks of the mailboxes (0: never)


def p2p_check(group=None, what="a training step"):
    """Collective health check of the peer-to-peer transport (train_step.TrainStep calls it every CHECK_EVERY steps and at capture):
    every rank reads its error words, the ranks agree through a MAX all-reduce, and ALL of them raise if any exchange of any rank
    timed out or found an overwritten slot -- from that exchange on the channel returned NaN statistics, so the step's numbers are
    void; the run must not go on (RCCL would have blocked instead).  Returns quietly when everything is fine."""
    if not p2p_active():
        return
    bits = p2p_status()
    grp = _p2p["group"] if group is None else group
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(grp) > 1:
        flag = torch.tensor([bits], dtype=torch.int32, device="cuda" if dist.get_backend(grp) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=grp)
        bits = int(flag.item())
    if bits:
        raise RuntimeError(f"npp_amd.comm: the peer-to-peer SyncBatchNorm exchange failed during {what} (error bits {bits}: "
                           "1 = a peer did not answer within NPP_P2P_TIMEOUT_MS, 2 = a peer ran ahead and overwrote a mailbox slot); "
                           "the statistics of that exchange and everything after it are NaN.  Restart from the last checkpoint; "
                           "NPP_SYNCBN_P2P=0 keeps every exchange on RCCL collectives")
