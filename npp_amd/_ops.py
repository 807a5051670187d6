"""torch.autograd.Function wrappers around the C-ABI kernels (include/npp_hip.h).

Every tensor that crosses this layer is logical NCHW with NHWC memory (channel stride 1), possibly a
channel slice of a wider buffer.  PyTorch is used for allocation, the autograd tape and streams only:
every byte of activation / gradient arithmetic is done by libnpp_hip.  No fallbacks.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch.autograd import Function

from . import _lib as L
from ._lib import check, desc, geom, lib, new_nhwc, ptr, stream_ptr, to_nhwc, tref

BN_MOMENTUM = 0.1


R = L.STAT_REPLICAS


def _byref(t):
    return C.byref(desc(t))


# --------------------------------------------------------------------------------------------------
# phase stamps (diagnostics, NPP_STAMPS=1): GPU-side wall-clock marks at phase boundaries of the step, valid inside a captured graph
# --------------------------------------------------------------------------------------------------
STAMPS = None      # {"buf": int64 tensor, "names": [(name, raw stream handle)]} while collecting (tools/phase_stamps.py)


def stamps_begin(device, n=512):
    global STAMPS
    STAMPS = {"buf": torch.zeros(n, dtype=torch.int64, device=device), "names": []}


def stamp(name):
    st = STAMPS
    if st is None or len(st["names"]) >= st["buf"].numel():
        return
    idx = len(st["names"])
    st["names"].append((name, stream_ptr()))
    check(lib().npp_stamp(st["buf"].data_ptr(), idx, stream_ptr()), "npp_stamp")


class _StampThrough(Function):
    @staticmethod
    def forward(ctx, x, name):
        ctx.name = name
        stamp(name + " fwd")
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        stamp(ctx.name + " bwd")
        return g, None


def stamp_through(x, name):
    """Identity on `x` that stamps `name fwd` now and `name bwd` when the gradient passes (on whatever stream autograd runs it)."""
    if STAMPS is None or not isinstance(x, torch.Tensor):
        return x
    return _StampThrough.apply(x, name)


# --------------------------------------------------------------------------------------------------
# fan-out: one-pass gradient accumulation for tensors with several consumers
# --------------------------------------------------------------------------------------------------
# A cell state feeds up to four ops (model_augment.py:48-62, genotypes.py:30-54); the autograd engine sums their gradients
# with a chain of binary at::add launches (measured 1.5 TB/s on MI355X: 364 launches, 3.3 ms per step at batch 16).  Every
# op wrapper below therefore consumes its input through take(): the first consumer turns the tensor into the input of ONE
# _FanOut node with several aliasing outputs, each later consumer receives the next alias, and _FanOut.backward adds all
# incoming gradients in one npp_add_n launch (n reads + 1 write).  NPP_FANOUT=0 restores the engine's accumulation.
FANOUT = os.environ.get("NPP_FANOUT", "1") != "0"
_FAN_N = 6                       # aliases per node (a cell state feeds up to 4 ops + the concatenation); the last one is reserved to chain a further node when they run out
_FAN_REG: dict = {}              # id(tensor) -> [aliases, handed out, weakref]; dropped by fan_reset() (Network.forward)


def add_n(ts: Sequence[torch.Tensor]) -> torch.Tensor:
    """ts[0] + ... + ts[n-1] in one launch (no autograd)."""
    ts = [to_nhwc(t) for t in ts]
    dt = ts[0].dtype
    ts = [t if t.dtype == dt else cast(t, dt) for t in ts]
    out = new_nhwc(*ts[0].shape, dt, ts[0].device)
    for lo in range(0, len(ts), 7):            # > 8 sources (never in this network): fold 7 at a time into the output
        part = ts[lo:lo + 7] if lo == 0 else [out] + ts[lo:lo + 7]
        if lo == 0 and len(ts) <= 8:
            part = ts
        descs = [desc(t) for t in part]
        arr = (C.POINTER(L.NppTensor) * len(part))(*[C.pointer(d) for d in descs])
        check(lib().npp_add_n(arr, len(part), _byref(out), stream_ptr()), "npp_add_n")
        if len(ts) <= 8:
            break
    return out


# Gradient accumulation IN the producers (round 3).  The consumers of a fan-out tensor whose data gradient comes
# out of an LDS-DMA conv kernel (conv_g4 / conv_h3 / conv_g8 / conv_thin: the preprocess and head convs of the 96 x 96 maps, where
# the bytes are) share ONE gradient buffer: the first of them stores, the later ones read-add-store in their epilogue (NppConvGeom
# relu_in bit 1), all on the stream of the first -- the add_n pass over n + 1 tensors (or n of its n + 1) disappears.  Other
# consumers (pooling, depthwise, SE, plain adds) still hand back tensors of their own and _FanOut.backward adds what is left.
FAN_ACCUM = os.environ.get("NPP_FAN_ACCUM", "1") != "0"
FAN_STATS = [0, 0, 0]      # claims that stored / accumulated / fell back to a private tensor


class _Like:
    """shape / dtype / device of a tensor that is not at hand (a backward that saved only the shape of its input)."""
    __slots__ = ("shape", "dtype", "device")

    def __init__(self, shape, dtype, device):
        self.shape, self.dtype, self.device = torch.Size(shape), dtype, device


# NPP_DBG_FAN_CENSUS=1: which backward is the LAST writer of a fan-out tensor's gradient ("next" item 1)
FAN_CENSUS = __import__("collections").Counter() if os.environ.get("NPP_DBG_FAN_CENSUS") else None
if FAN_CENSUS is not None:
    import atexit
    atexit.register(lambda: [print(f"npp-fan {n:6d} {k}", file=__import__("sys").stderr) for k, n in FAN_CENSUS.most_common()])


# BatchNorm-backward sums in the LAST writer of the gradient (npp_conv_dgrad_sums, csrc/conv_epi.h SUMS): the output T of a fused local
# BatchNorm add carries a ticket (its raw inputs, their mean | invstd); the consumers of T were counted in forward (take()), so the data
# gradient whose claim completes that count knows it finishes the gradient of T and adds sum g, sum g * xhat_a [, sum g * xhat_b] into the
# slabs the BatchNorm backward's apply kernel reads -- the stand-alone npp_bn_bwd_reduce* launch (on the dependent chain, ~15 us of wall
# time each) is skipped.  NPP_BN_SUMS=0: every BatchNorm backward reduces for itself.
BN_SUMS = os.environ.get("NPP_BN_SUMS", "1") != "0"
BN_SUMS_STATS = [0, 0, 0]      # tickets issued / sums delivered by a data gradient / delivered sums consumed by a BatchNorm backward


class _BnTicket:
    __slots__ = ("two", "ya", "yb", "mia", "mib", "c", "sums", "buf_ptr", "buf", "buf_ver")

    def __init__(self, ya, yb, mia, mib):
        self.two = yb is not None
        self.ya, self.yb, self.mia, self.mib, self.c = ya, yb, mia, mib, ya.shape[1]
        self.sums, self.buf_ptr, self.buf, self.buf_ver = None, 0, None, 0

    def deliver(self, sums, buf):
        """`sums` belong to the gradient held in `buf` as it stands now.  The ticket keeps `buf` alive (a second owner: the autograd
        engine's InputBuffer never accumulates another gradient into it in place) and remembers its version counter -- an in-place
        add by anybody else (`add_`) would bump it; this library's kernels write through raw pointers and do not."""
        self.sums, self.buf_ptr, self.buf, self.buf_ver = sums, buf.data_ptr(), buf, buf._version

    def drop(self):
        self.sums, self.buf_ptr, self.buf, self.buf_ver = None, 0, None, 0

    def take(self, dout, two):
        """The delivered sums if they belong to `dout` (and to this kind of backward), once."""
        sums, ptr_, buf, ver = self.sums, self.buf_ptr, self.buf, self.buf_ver
        self.drop()
        if (sums is None or buf is None or ptr_ != dout.data_ptr() or two != self.two or dout.shape != buf.shape
                or dout.stride() != buf.stride() or buf._version != ver or dout._version != ver):
            return None
        BN_SUMS_STATS[2] += 1
        return sums


class _FanAcc:
    """Shared data-gradient buffer of one fan-out tensor."""
    __slots__ = ("buf", "stream", "last", "ticket", "st", "writers", "broken")

    def __init__(self):
        self.buf, self.stream, self.last = None, None, None
        self.ticket, self.st, self.writers, self.broken = None, None, 0, False

    def last_writer_ticket(self):
        """The ticket of the fan-out tensor if the claim just made was the LAST expected one: every consumer counted in forward has now
        written into the shared buffer on this stream (a consumer that could not is `broken`)."""
        tk = self.ticket
        if tk is None or self.broken or self.st is None or self.buf is None:
            return None
        k = self.st[1]
        if k >= _FAN_N - 1 or self.writers != k:      # (a chained fan-out node: consumers this node does not count)
            return None
        return tk

    def seed(self, grad):
        """`grad` (an OWNED tensor nobody else reads: a channel slice of a concatenation's gradient) becomes the shared buffer: the
        consumers that come later add into it, and the slice needs no add_n pass of its own."""
        if self.buf is None and grad.dtype == torch.bfloat16:
            self.buf, self.stream = grad, stream_ptr()
            self.writers += 1
            FAN_STATS[0] += 1
        else:
            self.broken = True

    def claim(self, like):
        """(buffer, accumulate?) for a consumer about to write the gradient of `like` on the current stream, or (None, False)."""
        if FAN_CENSUS is not None:
            import traceback
            fr = traceback.extract_stack(limit=4)
            self.last = "/".join(f"{f.name}:{f.lineno}" for f in fr[:-1][-2:])      # (the backward that claims: census of the LAST writer)
        cur = stream_ptr()
        if self.buf is None:
            self.buf = new_nhwc(*like.shape, like.dtype, like.device)
            self.stream = cur
            self.writers += 1
            FAN_STATS[0] += 1
            return self.buf, False
        if cur == self.stream and self.buf.dtype == like.dtype and self.buf.shape == like.shape:
            self.writers += 1
            FAN_STATS[1] += 1
            return self.buf, True
        self.broken = True
        FAN_STATS[2] += 1
        return None, False


def _claim_dx(fan, like):
    """(dx, accumulate?) for a backward about to write the gradient of `like`: the fan-out node's shared buffer when there is one
    (first claimant stores, later ones add), else a tensor of its own."""
    if fan is not None:
        dx, accumulate = fan.claim(like)
        if dx is not None:
            return dx, accumulate
    return new_nhwc(*like.shape, like.dtype, like.device), False


def _unclaim():
    """The kernel behind an accumulating claim cannot add into its output: the claim is replaced by a private tensor."""
    FAN_STATS[1] -= 1
    FAN_STATS[2] += 1


def take_acc(x):
    """take(x) plus the shared gradient buffer (_FanAcc) of x's fan-out node -- None when x needs no fan-out node or accumulation
    is off.  For consumers whose data-gradient kernel can add into an existing tensor."""
    a = take(x)
    if not FAN_ACCUM or a is x or not x.is_cuda or x.dtype != torch.bfloat16:
        return a, None
    st = _FAN_REG.get(id(x))
    if st is None or st[2]() is not x or not any(a is o for o in st[0][:-1]):
        return a, None           # (the alias came from a chained node -- a 4th, 5th ... consumer: a tensor of its own, or the chained
                                 #  node's sum would contain the shared buffer a second time)
    if not st[3]:
        fa = _FanAcc()
        fa.ticket, fa.st = st[4], st
        st[3].append(fa)
    return a, st[3][0]


def _mark_owned(t):
    """`t` is a gradient tensor this layer made for ONE reader (an add_n result, a finished shared buffer): whoever receives it may
    add into it in place.  Pass-through gradients (a plain add hands the SAME tensor to both operands) never carry the mark."""
    try:
        t._npp_own = True
    except Exception:      # noqa: BLE001
        pass


class _FanOut(Function):
    @staticmethod
    def forward(ctx, x, holder):
        ctx.set_materialize_grads(False)       # aliases nobody consumed hand back None, not a zero tensor
        ctx.holder = holder                    # [ _FanAcc ] once a consumer asked for the shared gradient buffer (take_acc)
        ctx.is_bn = bool(getattr(x, "_npp_bn", False))
        return tuple(x.view_as(x) for _ in range(_FAN_N))

    @staticmethod
    def backward(ctx, *gs):
        acc = ctx.holder[0] if ctx.holder else None
        shared = acc.buf if acc is not None else None
        uniq = []
        seen_shared = False
        for g in gs:
            if g is None:
                continue
            # consumers that accumulated all hand back the ONE shared buffer: it counts once (compared by memory, the engine need
            # not hand back the same Python object); every other gradient is a contribution of its own, equal tensors included
            if shared is not None and g.data_ptr() == shared.data_ptr() and g.shape == shared.shape and g.stride() == shared.stride():
                if seen_shared:
                    continue
                seen_shared = True
            uniq.append(g)
        if FAN_CENSUS is not None and uniq:
            kind = "add_n" if len(uniq) > 1 else (("shared:" + str(acc.last)) if seen_shared else "single:" + str(getattr(uniq[0], "_npp_prod", "?")))
            FAN_CENSUS[("bn" if ctx.is_bn else "other", kind, len(uniq))] += 1
        if acc is not None:
            if acc.ticket is not None and acc.ticket.sums is not None and not (len(uniq) == 1 and seen_shared):
                acc.ticket.drop()      # (somebody's gradient came in beside the shared buffer)
            acc.buf = acc.stream = None        # a later backward over the same graph (retain_graph) starts a new buffer
            acc.writers, acc.broken = 0, False
        if not uniq:
            return None, None
        if len(uniq) == 1:
            if seen_shared:
                _mark_owned(uniq[0])
            return uniq[0], None
        r = add_n(uniq)
        _mark_owned(r)
        return r, None


def take(x):
    """The tensor a consumer op should read instead of `x` (same memory): see the fan-out note above."""
    if (not FANOUT or not isinstance(x, torch.Tensor) or x.dim() != 4 or not torch.is_grad_enabled()
            or not x.requires_grad or x.grad_fn is None):
        return x
    st = _FAN_REG.get(id(x))
    if st is None or st[2]() is not x:         # (a dead Python wrapper's id may be reused by another tensor)
        if len(_FAN_REG) > 8192:               # ops used outside a Network.forward: never grow without bound
            _FAN_REG.clear()
        holder = []
        st = [_FanOut.apply(x, holder), 0, weakref.ref(x), holder, getattr(x, "_npp_ticket", None)]
        _FAN_REG[id(x)] = st
    outs, k = st[0], st[1]
    if k < _FAN_N - 1:
        st[1] = k + 1
        return outs[k]
    return take(outs[-1])                      # out of aliases: the reserved one becomes the input of a further node


def fan_reset():
    """Forget the aliases handed out so far (they keep their base tensors alive); called around every Network.forward."""
    _FAN_REG.clear()
    _RELU_MASKS.clear()


# --------------------------------------------------------------------------------------------------
# ReLU bit-masks: 1 bit per element, written by the kernel that produces a tensor, read by the data gradient of a
# `ReLU -> conv` consumer instead of the bf16 tensor itself (1/16 of the bytes; every op starts with nn.ReLU, operations.py:69-82)
# --------------------------------------------------------------------------------------------------
RELU_BITS = os.environ.get("NPP_RELU_BITS", "1") != "0"
MASK_STATS = [0, 0, 0]      # data gradients that read a bit-mask / found one but ran on a kernel without support / found none
_RELU_MASKS: dict = {}      # (data_ptr, channels) -> (weakref to the produced tensor, mask bytes, byte offset, bytes per pixel)


def _mask_wanted(like: torch.Tensor) -> bool:
    return (RELU_BITS and like.dtype == torch.bfloat16 and like.dim() == 4 and like.shape[1] % 8 == 0
            and torch.is_grad_enabled())


def _register_mask(t: torch.Tensor, spec):
    if spec is not None:
        if len(_RELU_MASKS) > 4096:
            _RELU_MASKS.clear()
        _RELU_MASKS[(t.data_ptr(), t.shape[1])] = (weakref.ref(t), spec[0], spec[1], spec[2])


def relu_mask_of(x: torch.Tensor):
    """NppTensor descriptor (+ the tensor that owns the bytes) of the bit-mask of `x`, if its producer wrote one."""
    ent = _RELU_MASKS.get((x.data_ptr(), x.shape[1]))      # (part 0 of a concatenation starts where the whole does)
    if ent is None:
        return None
    t = ent[0]()
    # the producer's tensor object must still be alive (else the address may belong to someone else by now) and be this tensor
    if t is None or t.data_ptr() != x.data_ptr() or t.shape != x.shape or t.stride() != x.stride():
        return None
    n, c, h, w = x.shape
    return L.NppTensor(ent[1].data_ptr() + ent[2], n, h, w, c, ent[3], L.NPP_MASK8, 0), ent[1]


class _ZeroPool:
    """Pre-zeroed scratch carved from large chunks: one memset per chunk instead of one fill launch per
    statistics / gradient accumulator (~3000 tiny fills per training step otherwise)."""

    def __init__(self, dtype, chunk_elems):
        self.dtype, self.chunk = dtype, chunk_elems
        self.buf, self.off = None, 0

    def get(self, n: int, device, own: bool = False) -> torch.Tensor:
        n_al = (n + 63) // 64 * 64
        if self.buf is None or self.buf.device != device or self.off + n_al > self.buf.numel():
            self.buf = torch.zeros(max(self.chunk, n_al), dtype=self.dtype, device=device)
            self.off = 0
        if not own:
            t = self.buf[self.off:self.off + n]
            self.off += n_al
            return t
        # own=True -- not a view: a tensor of its own on the chunk's storage.  Slices handed out here end up saved for backward
        # (SE squeeze) AND as parameter gradients (1x1 weight gradients); views of one buffer share ONE version counter, so an
        # in-place `p.grad += g` (a second backward before zero_grad, as the alpha pass of train_with_alpha does,
        # core/function.py:613-615) would invalidate every saved slice of the chunk
        t = torch.empty(0, dtype=self.dtype, device=self.buf.device).set_(self.buf.untyped_storage(), self.off, (n,), (1,))
        self.off += n_al
        return t


_HAVE_GPU = None


def _have_gpu() -> bool:
    global _HAVE_GPU
    if _HAVE_GPU is None:
        _HAVE_GPU = bool(torch.cuda.is_available())
    return _HAVE_GPU


class _PerStream:
    """One pool per HIP stream: a chunk is zeroed by a fill on the stream that creates it, so its slices may only be
    handed to kernels of that stream (the network runs its two branches on two streams)."""

    def __init__(self, make):
        self.make, self.pools = make, {}

    def cur(self):
        key = stream_ptr() if _have_gpu() else 0
        pool = self.pools.get(key)
        if pool is None:
            pool = self.pools[key] = self.make()
        return pool

    def all(self):
        return list(self.pools.values())


GRAPH_TOPOLOGY = False  # set by train_step.TrainStep while its steps are (to be) replayed as a hipGraph: Network.forward then
                        # picks the stream topology that pays off under replay (SyncBN: hub streams + lockstep issue)
_hub_stream = None     # the stream Network.forward was called on while it runs its branches on two streams


BN_PAIRS = os.environ.get("NPP_BN_PAIRS", "1") != "0"      # under SYNC_WAVES: pairs of fused BatchNorm adds as one autograd node (_BnAddPair)
SYNC_WAVES = False     # set by Network.forward under SyncBatchNorm: cells issue the edges of every ready node before any BatchNorm apply
P2P_DIRECT = False     # set by Network.forward: the SyncBatchNorm exchanges go through the peer-to-peer mailboxes (csrc/p2p.hip), each on
                       # the stream of the kernels that need it -- no hub stream, no lockstep (an exchange is an ordinary kernel)


def _sum_all_reduce(t, group):
    """SUM all-reduce on the current stream: the library's RCCL communicator when it is up (npp_amd.comm), else torch's."""
    from . import comm
    if t.is_cuda and not comm.p2p_active():
        comm.ensure_p2p(group)      # first exchange of this group: every rank is at the same point of the program
    if comm.p2p_active() and comm.p2p_exchange(t, group):      # one-shot peer-to-peer exchange (csrc/p2p.hip), one node
        return
    if comm.active() and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and group is comm._state["group"]:
        comm.syncbn_exchange(t)
    else:
        dist.all_reduce(t, group=group)


def hub_all_reduce(t, group, producers=None):
    """SyncBatchNorm exchange.  Every collective of the step is issued on ONE stream (the caller's / the hipGraph
    capture's origin stream): a branch running on the side stream hands over with an event each way.  One stream per
    communicator is the topology RCCL-in-hipGraph users run; and the only cross-stream edges this adds to a capture
    involve its origin stream (see helper_stream for what ROCm 7.0 does otherwise).
    producers: further streams whose kernels wrote into `t` (a merged exchange of both branches): the hub waits for them too.
    Returns the event recorded on the hub after the collective (None when everything ran on the current stream)."""
    hub = _hub_stream
    cur = torch.cuda.current_stream() if t.is_cuda else None
    if hub is None or cur is None:
        _sum_all_reduce(t, group)
        return None
    if P2P_DIRECT and not producers:
        from . import comm
        if comm.p2p_active() and comm.p2p_exchange(t, group):      # an ordinary kernel on the stream that needs the sums
            return None
        # (no mailbox channel left for this stream: the collective, on the hub stream as always)
    same = cur.cuda_stream == hub.cuda_stream
    waited = {hub.cuda_stream}
    for st in ([] if same else [cur]) + list(producers or ()):
        if st is not None and st.cuda_stream not in waited:
            waited.add(st.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(st)
            hub.wait_event(ev)
    if same and not producers:
        _sum_all_reduce(t, group)
        return None
    with torch.cuda.stream(hub):
        _sum_all_reduce(t, group)
        back = torch.cuda.Event()
        back.record(hub)
    if not same:
        cur.wait_event(back)
    return back


_helper_streams = {}
_branch_b_streams = set()      # raw handles of the side streams the network runs its second branch on
_helper_uses = 0


_hub_offload = None      # (hub stream, {raw handles of the branch streams}) while Network.forward runs the hub topology
_hub_counter = 0


def helper_stream_for_edge():
    """Stream for the second edge of a cell node: the hub stream for a share of the nodes of either branch (hub topology),
    else the eager helper stream (NPP_STREAMS=4), else None."""
    global _hub_counter
    off = _hub_offload
    if off is not None:
        cur = torch.cuda.current_stream()
        if cur.cuda_stream in off[1]:
            _hub_counter += 1
            num, den = _HUB_SHARE
            if (_hub_counter % den) < num:
                return off[0]
        return None
    return helper_stream()


def _parse_share():
    v = os.environ.get("NPP_HUB_SHARE", "2/3")
    try:
        a, b = v.split("/")
        return max(0, int(a)), max(1, int(b))
    except ValueError:
        return 2, 3


_HUB_SHARE = _parse_share()


def helper_stream():
    """The helper stream paired with the current stream (NPP_STREAMS=4: the two edges of a cell node overlap), or None.
    Opt-in and eager-only: on ROCm 7.0 hipStreamEndCapture segfaults when two non-origin streams of a capture wait on
    each other in both directions (tools/capture_repro.py, variants `bidir` / `nested`), which is exactly what the
    helper of the side-stream branch does.  Fork/join with the capture's origin stream is fine, hence 2 streams."""
    if os.environ.get("NPP_STREAMS", "2") != "4" or not _have_gpu():
        return None
    if torch.cuda.is_current_stream_capturing():
        return None
    global _helper_uses
    lim = os.environ.get("NPP_HELPER_MAX")
    if lim is not None:
        _helper_uses += 1
        if _helper_uses > int(lim):
            return None
    # two helpers per device, created together and before any hipGraph capture (streams created while a capture is in
    # progress crash hipStreamEndCapture on this stack): [0] serves whatever stream the caller runs on, [1] the
    # parsing-branch stream registered by the network
    cur = torch.cuda.current_stream()
    key = (cur.device.type, cur.device.index)
    pair = _helper_streams.get(key)
    if pair is None:
        pair = _helper_streams[key] = (torch.cuda.Stream(device=cur.device), torch.cuda.Stream(device=cur.device))
    return pair[1] if cur.cuda_stream in _branch_b_streams else pair[0]


_zpool64 = _PerStream(lambda: _ZeroPool(torch.float64, 1 << 21))
_zpool32 = _PerStream(lambda: _ZeroPool(torch.float32, 1 << 24))


_stream_caches: list = []      # dict / set objects other modules cache HIP streams in (model_augment._side_streams)


def join_capturing_side_streams():
    """Inside a capture: make the current (origin) stream wait for every cached side stream that has joined the capture.  A
    backward run by torch.autograd.grad ends on whatever streams its last nodes ran on (the engine joins leaf streams only for
    AccumulateGrad nodes); ending the capture with a forked stream unjoined is an error (a crash on this ROCm)."""
    cur = torch.cuda.current_stream()
    seen = {cur.cuda_stream}
    streams = [st for pair in _helper_streams.values() for st in (pair if isinstance(pair, (tuple, list)) else (pair,))]
    for c in _stream_caches:
        streams += list(c.values()) if isinstance(c, dict) else list(c)
    for st in streams:
        if not isinstance(st, torch.cuda.Stream) or st.cuda_stream in seen:
            continue
        seen.add(st.cuda_stream)
        with torch.cuda.stream(st):
            capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            cur.wait_stream(st)


def note_stream_use(t, stream):
    """`t` (allocated on some other stream) is about to be read or written by kernels of `stream`: tell the caching allocator, so that
    the block is not reused before that stream's work is done (Tensor.record_stream; a no-op on the CPU)."""
    if t is not None and t.is_cuda and stream is not None:
        t.record_stream(stream)


def forget_streams():
    """Drop every cached side stream.  After a hipGraph capture that was invalidated half way the streams that had joined it
    stay in capture mode for good on this ROCm (hipStreamEndCapture fails on them): the eager fallback needs fresh ones."""
    _helper_streams.clear()
    _branch_b_streams.clear()
    for c in _stream_caches:
        c.clear()
    for ps in (_zpool64, _zpool32, _sync_pool):
        ps.pools.clear()
    _shared_sync_pool.drop()


def reset_pools():
    """Drop the current scratch chunks (call after CUDA-graph capture: captured chunks belong to the graph)."""
    for pool in _zpool64.all() + _zpool32.all():
        pool.buf = None
    for pool in _sync_pool.all():
        pool.drop()


def zeros_f64(n, device):
    return _zpool64.cur().get(n, device)


class _SyncStatsPool(_ZeroPool):
    """Statistics of SyncBatchNorm layers live back to back in one chunk, so that every vector produced since the
    last exchange travels in ONE all-reduce of the range [synced_off, off) -- no gather / concat / slice kernels
    around the collective (augment_lip_sync.py:191 turns all 490 BNs into SyncBatchNorm).

    `shared`: ONE pool for both task-branch streams (lockstep issue under the hub topology, see SYNC_MERGE): the chunk's zero-fill
    is fenced with an event every other stream waits for before its first kernel writes into the chunk, a flush makes the hub
    wait for EVERY stream that produced a waiting vector, and a side that finds itself already exchanged by the other branch's
    flush waits for that flush's completion event."""

    def __init__(self, chunk_elems, shared=False):
        super().__init__(torch.float64, chunk_elems)
        self.synced_off = 0
        self.waiting = []      # BnSides whose statistics sit in the un-exchanged range
        self.group = None
        self.ws = 1
        self.shared = shared
        self.fill_ev = None    # zero-fill of the current chunk (shared pools)
        self.fill_seen = set() # raw handles of the streams ordered behind it
        self.flushes = 0

    def get(self, n, device):
        n_al = (n + 63) // 64 * 64
        if self.buf is not None and self.off + n_al > self.buf.numel():
            self.flush()       # the old chunk is about to be left behind
        if self.buf is None or self.buf.device != device or self.off + n_al > self.buf.numel():
            self.buf = torch.zeros(max(self.chunk, n_al), dtype=self.dtype, device=device)
            self.off = self.synced_off = 0
            if self.shared and device.type == "cuda":
                cur = torch.cuda.current_stream()
                self.fill_ev = torch.cuda.Event()
                self.fill_ev.record(cur)
                self.fill_seen = {cur.cuda_stream}
                for st in _known_streams():
                    self.buf.record_stream(st)
        if self.shared and self.fill_ev is not None:
            cur = torch.cuda.current_stream()
            if cur.cuda_stream not in self.fill_seen:
                cur.wait_event(self.fill_ev)
                self.fill_seen.add(cur.cuda_stream)
        return super().get(n, device)

    def holds_unsynced(self, t):
        if self.buf is None or t is None or t.device != self.buf.device:
            return False
        d = t.data_ptr() - self.buf.data_ptr()
        return self.synced_off * 8 <= d < self.off * 8

    def enlist(self, side, grp, ws):
        self.waiting.append(side)
        side.stream = torch.cuda.current_stream() if (self.shared and side.x.is_cuda) else None
        self.group, self.ws = grp, ws

    def flush(self):
        if self.waiting:
            rng = self.buf[self.synced_off:self.off]
            if self.shared:
                back = hub_all_reduce(rng, self.group, producers=[sd.stream for sd in self.waiting if sd.stream is not None])
                for sd in self.waiting:
                    sd.sync_event = back
            else:
                if not self._flush_slabs():
                    if P2P_FOLD and P2P_DIRECT:
                        # (never the pool's range in place while exchanges are folded into kernels: rows of a merged conv may be
                        #  read by a consumer on the other branch stream at this very moment -- every vector on its own)
                        for sd in self.waiting:
                            if sd.stats_c:
                                _compact_stats(sd)
                            hub_all_reduce(sd.stats, self.group)
                    else:
                        hub_all_reduce(rng, self.group)
            self.flushes += 1
            for sd in self.waiting:
                sd.synced_ws = self.ws
            self.waiting = []
        self.synced_off = self.off

    def _flush_slabs(self) -> bool:
        """Peer-to-peer transport: every waiting BatchNorm's [R][2C] replica slabs are collapsed by the exchange kernel itself and
        only the 2C sums travel (1/16 of the bytes of exchanging the pool's range as it is); the world's sums land in replica 0, the
        other replicas are zeroed -- the consumers (npp_affine_add_fin, npp_bn_finalize) sum all replicas as before."""
        if not P2P_DIRECT or _hub_stream is None:
            return False
        from . import comm
        if not comm.p2p_active():
            return False
        segs = []
        for sd in self.waiting:
            if P2P_FOLD and sd.stats_c:
                # With exchanges folded into the consuming kernels, some edges of a merged conv may already have traded their slice
                # of the conv's statistics rows -- possibly on the other branch stream, possibly right now.  The rows are never
                # exchanged in place then: every edge still waiting here travels as a private copy of its own slice.
                _compact_stats(sd)
                sd.rider = False
                sd.carry = None
            if sd.rider:
                continue
            st = sd.carry if sd.carry is not None else sd.stats
            if st is None or not st.is_cuda or st.dtype != torch.float64 or not st.is_contiguous() or st.numel() % R != 0:
                return False
            ln = st.numel() // R
            segs.append((st, ln, R, ln, (None, None, None, None), True))
        # pieces of at most 8 segments AND at most one mailbox slot of doubles each, all validated before the first one is sent: the
        # decision "mailboxes or collective" is made up front, never in the middle of a flush
        if not segs:
            return False
        cap = comm._p2p["cap"]
        pieces, cur, tot = [], [], 0
        for sg in segs:
            if sg[1] > cap:
                return False
            if cur and (len(cur) == 8 or tot + sg[1] > cap):
                pieces.append(cur)
                cur, tot = [], 0
            cur.append(sg)
            tot += sg[1]
        pieces.append(cur)
        if not all(comm.p2p_can(sum(s_[1] for s_ in pc), self.group) for pc in pieces):
            return False
        for pc in pieces:
            if not comm.p2p_exchange_slabs(pc, self.group):
                raise RuntimeError("peer-to-peer exchange refused a piece it had accepted the size of")
        return True

    def drop(self):
        self.buf, self.off, self.synced_off, self.waiting = None, 0, 0, []
        self.fill_ev, self.fill_seen = None, set()


def _known_streams():
    sts = []
    if _hub_stream is not None:
        sts.append(_hub_stream)
    for c in _stream_caches:
        sts.extend(v for v in c.values() if isinstance(v, torch.cuda.Stream))
    return sts


# SyncBatchNorm under the lockstep issue order (model_augment._lockstep): the two branches' statistics share one pool, so the
# exchange a branch triggers also carries what the OTHER branch has produced since the last one -- one collective per lockstep
# stage instead of one per branch and stage in the forward pass (~490 -> ~245 forward exchanges per step; the backward exchanges
# stay per node: a node's backward needs its all-reduced sums before the host can even start the other branch's node).
SYNC_MERGE = False
_shared_sync_pool = _SyncStatsPool(1 << 21, shared=True)


class _SyncPools(_PerStream):
    def cur(self):
        return _shared_sync_pool if SYNC_MERGE else super().cur()

    def all(self):
        return super().all() + [_shared_sync_pool]


_sync_pool = _SyncPools(lambda: _SyncStatsPool(1 << 21))


def stats_buffer(n, device, want_stats):
    """want_stats: 1 = local statistics, 2 = statistics that will be exchanged across ranks (SyncBatchNorm)."""
    return _sync_pool.cur().get(n, device) if want_stats == 2 else _zpool64.cur().get(n, device)


def zeros_f32(n, device, own=False):
    """own: the slice will be saved for backward or become a parameter gradient (see _ZeroPool.get)."""
    return _zpool32.cur().get(n, device, own)


# --------------------------------------------------------------------------------------------------
# packed-weight cache: f32 OIHW parameter -> MFMA operand image, rebuilt when the parameter changes
# --------------------------------------------------------------------------------------------------
_pack_cache = {}   # id(parameter) -> (weakref, {(for_dgrad, dtype): ((version, data_ptr), packed)})
_pack_by_ptr = {}  # data_ptr -> id(parameter) of the WeightPacker-managed entries (lookup for aliases of a parameter)


def packed_weight(w: torch.Tensor, for_dgrad: bool, dtype: torch.dtype) -> torch.Tensor:
    key = (bool(for_dgrad), dtype)
    ver = (w._version, w.data_ptr())
    wid = id(w)
    ent = _pack_cache.get(wid)
    if ent is None or ent[0]() is not w:
        # an ALIAS of a parameter (auto_graph captures gradients w.r.t. `p.view_as(p)`): same storage, same version counter --
        # the images the owner's entry holds are the alias's images
        owner = _pack_cache.get(_pack_by_ptr.get(w.data_ptr(), 0))
        o = owner[0]() if owner is not None else None
        if o is not None and o.data_ptr() == w.data_ptr() and o.shape == w.shape and o._version == w._version:
            ent = owner
        else:
            ent = (weakref.ref(w, lambda _r, _k=wid: _pack_cache.pop(_k, None)), {})
            _pack_cache[wid] = ent
    per = ent[1]
    hit = per.get(key)
    if hit is not None and hit[0] == ver:
        # A version counter is not proof of freshness (torch.optim.Adam(fused=True) updates without bumping it).  Images
        # written by a model's WeightPacker in this forward are marked managed; a free-standing op in a training
        # forward repacks, and drops its data-gradient image so that backward repacks too.
        if len(hit) > 2 or for_dgrad or not (torch.is_grad_enabled() and w.requires_grad):
            return hit[1]
        per.pop((True, dtype), None)
    co, ci, kh, kw = w.shape
    n = lib().npp_packed_weight_elems(co, ci, kh, kw, int(for_dgrad))
    out = torch.empty(n, dtype=dtype, device=w.device)
    wf = w.detach()
    if wf.dtype != torch.float32 or not wf.is_contiguous():
        wf = wf.float().contiguous()
    check(lib().npp_pack_weight(wf.data_ptr(), co, ci, kh, kw, int(for_dgrad), L.npp_dtype(dtype), out.data_ptr(),
                                stream_ptr()), "npp_pack_weight")
    per[key] = (ver, out)
    return out


def clear_caches():
    _pack_cache.clear()
    _pack_by_ptr.clear()


# --------------------------------------------------------------------------------------------------
# gradient slots: where a parameter's gradient is to be written (ddp.GradReducer's pre-flattened buckets)
# --------------------------------------------------------------------------------------------------
_GRAD_SLOTS: dict = {}     # parameter data_ptr -> (provider(param, zero) -> tensor or None, weakref to the parameter)


def register_grad_slot(p: torch.Tensor, provider):
    _GRAD_SLOTS[p.data_ptr()] = (provider, weakref.ref(p))


def unregister_grad_slot(key):
    _GRAD_SLOTS.pop(key, None)


def grad_out(param, shape=None, zero=False):
    """The tensor a backward kernel should write the f32 gradient of `param` into: its slot in a gradient bucket when a
    reducer has registered one and hands it out for this step (zero: the kernel accumulates, the slot must hold zeros --
    GradReducer.begin_step zeroed the bucket), else None (the caller allocates)."""
    if not _GRAD_SLOTS or param is None or param.dtype != torch.float32:
        return None
    ent = _GRAD_SLOTS.get(param.data_ptr())
    if ent is None:
        return None
    p = ent[1]()
    if p is None or p.shape != param.shape:
        return None
    return ent[0](p, zero)


def _grad_buf(param, n, device):
    """grad_out(param) or a fresh f32 vector of n elements."""
    t = grad_out(param)
    return t if t is not None else torch.empty(n, dtype=torch.float32, device=device)


TRAIN_EPOCH = 0    # bumped by every training forward of a Network and by every TrainStep replay: "the parameters may have been
                   # updated since" -- an optimizer can do that without moving Tensor._version (torch.optim.Adam(fused=True)
                   # never does; FusedAdam's increment_version ran once, at capture, not at the replays)


def note_training_step():
    global TRAIN_EPOCH
    TRAIN_EPOCH += 1


class WeightPacker:
    """Packs every dense-conv weight of a model (forward and data-gradient operand images) with ONE kernel
    launch whenever any of them changed (i.e. once per optimizer step) instead of ~820 tiny launches."""

    def __init__(self, weights, groups=()):
        self.weights = list(weights)
        self.groups = list(groups)      # WideGroups: merged images (conv2d_wide)
        self.sig = None
        self.epoch = -1            # TRAIN_EPOCH the images were last written at
        self.table = None
        self.outs = None
        self.key = None

    def _build(self, dtype, device):
        import numpy as np
        # merged edges: the group's forward image is its weights' own forward images back to back (rows = output channels), its
        # data-gradient image [Cin][tap][m * Cout] is filled column block by column block, one job per weight
        fwd_slice = {}
        self.group_bufs = []
        for g in self.groups:
            ws = g.weights
            _, ci, kh, kw = ws[0].shape
            run_bufs = {}
            for r, run in enumerate(g.runs):
                if len(run) < 2:
                    continue
                co = ws[run[0]].shape[0]
                n1 = int(lib().npp_packed_weight_elems(co, ci, kh, kw, 0))
                buf = torch.zeros(len(run) * n1, dtype=dtype, device=device)
                for q, k in enumerate(run):
                    fwd_slice[id(ws[k])] = buf[q * n1:(q + 1) * n1]
                run_bufs[r] = buf
            dgb = torch.zeros(int(lib().npp_packed_weight_elems(sum(g.cos), ci, kh, kw, 1)), dtype=dtype, device=device)
            self.group_bufs.append((g, run_bufs, dgb))
        njobs = 2 * len(self.weights) + sum(len(g.weights) for g in self.groups)
        jobs = (L.NppPackJob * njobs)()
        outs = []
        counts = []
        blk = 0
        i = 0
        for w in self.weights:
            co, ci, kh, kw = w.shape
            for dg in (0, 1):
                n = lib().npp_packed_weight_elems(co, ci, kh, kw, dg)
                out = fwd_slice.get(id(w)) if dg == 0 else None
                if out is None:
                    out = torch.zeros(n, dtype=dtype, device=device)      # padding stays zero: the kernel writes real elements only
                outs.append(out)
                jobs[i] = L.NppPackJob(w.data_ptr(), out.data_ptr(), co, ci, kh, kw, dg, L.npp_dtype(dtype), blk)
                nb = int(lib().npp_pack_job_blocks(co, ci, kh, kw, dg))
                counts.append(nb)
                blk += nb
                i += 1
        for g, _buf, dgb in self.group_bufs:
            ws = g.weights
            ct, off = sum(g.cos), 0
            for w in ws:
                co, ci, kh, kw = w.shape
                jobs[i] = L.NppPackJob(w.data_ptr(), dgb.data_ptr(), co, ci, kh, kw, 1, L.npp_dtype(dtype), blk, off, ct)
                nb = int(lib().npp_pack_job_blocks(co, ci, kh, kw, 1))
                counts.append(nb)
                blk += nb
                i += 1
                off += co
        assert i == njobs
        raw = np.frombuffer(bytes(jobs), dtype=np.uint8).copy()
        self.table = torch.from_numpy(raw).to(device)
        self.block_job = torch.from_numpy(np.repeat(np.arange(len(counts), dtype=np.int32), counts)).to(device)
        self.outs = outs
        self.njobs = njobs
        self.nblocks = blk
        self.key = (dtype, device, tuple(w.data_ptr() for w in self.weights))

    def pack_if_stale(self, dtype, device, force=False):
        """`force`: repack even if no version counter moved.  The network passes its `training` flag: an optimizer may
        update parameters without bumping `Tensor._version` (torch.optim.Adam(fused=True) does exactly that), and a
        training forward on stale operand images would silently use the previous step's weights."""
        ws = self.weights
        if not ws:
            return
        sig = tuple(w._version for w in ws)
        key = (dtype, device, tuple(w.data_ptr() for w in ws))
        if self.key != key:
            if any(w.dtype != torch.float32 or not w.is_contiguous() or w.device != device or w.shape[2] * w.shape[3] > 64 for w in ws):
                return        # unusual storage: leave it to the per-call path
            self._build(dtype, device)
            self.sig = None
        if sig == self.sig and not force and self.epoch == TRAIN_EPOCH:
            return         # (eval forward: no version moved AND no training step ran since the images were written)
        check(lib().npp_pack_weights_batched_map(self.table.data_ptr(), self.njobs, self.block_job.data_ptr(), self.nblocks,
                                                 stream_ptr()), "npp_pack_weights_batched_map")
        self.sig = sig
        self.epoch = TRAIN_EPOCH
        for g, run_bufs, dgb in self.group_bufs:
            g.img = {("d", dtype): dgb}
            for r, buf in run_bufs.items():
                g.img[("f", r, dtype)] = buf
        k = 0
        for w in ws:
            wid = id(w)
            ent = _pack_cache.get(wid)
            if ent is None or ent[0]() is not w:
                ent = (weakref.ref(w, lambda _r, _k=wid: _pack_cache.pop(_k, None)), {})
                _pack_cache[wid] = ent
            ver = (w._version, w.data_ptr())
            _pack_by_ptr[w.data_ptr()] = wid
            ent[1][(False, dtype)] = (ver, self.outs[k], True)
            ent[1][(True, dtype)] = (ver, self.outs[k + 1], True)
            k += 2


def _gemm_ready(x: torch.Tensor) -> torch.Tensor:
    """The MFMA gathers read whole 16-byte channel groups: a tensor whose channel count is not a multiple of 8
    must sit in rows padded (with zeros) to the next multiple.  Tensors made by this package already do."""
    c = x.shape[1]
    if c % 8 == 0:
        return x
    ld = L.nhwc_ld(x)
    cp = (c + 7) // 8 * 8
    if ld is not None and ld >= cp and ld % 8 == 0:
        # a channel slice of a wider buffer (one half of a concatenation's gradient): the group read of the LAST pixel must
        # still end inside the storage -- [4:8] of an 8-wide buffer would read 16 bytes past its end (a fault when the buffer
        # closes a mapped segment, NaN x 0 = NaN when the bytes happen to be a NaN pattern)
        n, _, h, w = x.shape
        if x.storage_offset() + (n * h * w - 1) * ld + cp <= x.untyped_storage().nbytes() // x.element_size():
            return x
    y = new_nhwc(*x.shape, x.dtype, x.device)
    check(lib().npp_copy(_byref(x), _byref(y), stream_ptr()), "npp_copy")
    return y


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _conv_out(h, k, s, p, d):
    return (h + 2 * p - d * (k - 1) - 1) // s + 1


# --------------------------------------------------------------------------------------------------
# dense conv (MFMA implicit GEMM)
# --------------------------------------------------------------------------------------------------
_conv_ws_cache = {}

# Deferred, batched unpack of the KxK weight gradients (packed [co][tap][ci] accumulation layout -> OIHW): a weight gradient has
# no reader before the optimizer, so a step that controls its own sequence (train_step.TrainStep without a reducer) lets the 224
# unpacks of a step wait and runs them as ONE launch (npp_unpack_wgrad_batched) between backward and optimizer.step().
DEFER_UNPACK = False
_pending_unpacks: list = []      # (src tensor, dst tensor, co, ci, taps, cp, kpad, nslabs, slab, stream)
_UNPACK_JOB = None


def _unpack_or_defer(src, dst, co, ci, kh, kw, nslabs, s, defer=None):
    taps = kh * kw
    cp = (ci + 7) // 8 * 8
    kpad = (taps * cp + 63) // 64 * 64
    if DEFER_UNPACK if defer is None else defer:
        # keep the destination's MEMORY alive through an alias with its own TensorImpl: a second reference to the tensor object
        # itself would make autograd's AccumulateGrad clone the (still unwritten) gradient instead of adopting it
        keep = torch.empty(0, dtype=dst.dtype, device=dst.device).set_(dst.untyped_storage(), dst.storage_offset(), dst.shape,
                                                                       dst.stride())
        _pending_unpacks.append((src, keep, co, ci, taps, cp, kpad, max(nslabs, 1), co * kpad, torch.cuda.current_stream()))
        return
    if nslabs > 0:
        check(lib().npp_unpack_wgrad_sum(src.data_ptr(), nslabs, co, ci, kh, kw, dst.data_ptr(), s), "npp_unpack_wgrad_sum")
    else:
        check(lib().npp_unpack_wgrad(src.data_ptr(), co, ci, kh, kw, dst.data_ptr(), s), "npp_unpack_wgrad")


# Deferred, batched weight gradients (npp_conv_wgrad_batched): nobody reads a weight gradient before the optimizer, and the small-map
# ones (12x12 / 24x24: ~100 blocks, ~25 us of latency each) are 140 launches and 2.9 ms of a step's kernel time.  Under TrainStep (no
# reducer) the weight gradients the LDS-DMA kernel can take are collected during backward -- x and dy stay alive, ~7 GB at batch 16 --
# and run as one launch per kernel variant before the batched unpack: off the two branch streams, and throughput- instead of
# latency-bound.  Measured (ms per step, same box) by pixel limit: 0 -> 50.9, 9 300 -> 49.8, 40 000 -> 49.7, 150 000 (all) -> 49.1-49.4.
DEFER_WGRAD_MAX_PIX = 0        # > 0: defer the weight gradients of maps with at most this many pixels (TrainStep sets it)
_pending_wgrads: list = []     # (x, dy, packed accumulator / gradient tensor, NppConvGeom, stream)
_wgrad_batchable: dict = {}
_wgrad_bufs = {"pin": None, "dev": None, "ev": None, "keep": []}
DEFER_TAIL_OK = True      # (test hook) False: TrainStep ignores GradReducer(overlap="tail") and runs the tail in one piece
_wgrad_bufs_k = {"pin": None, "dev": None, "ev": None, "keep": []}      # job tables of the KxK group (flush_wgrads(group="K"))


_deferred_params: set = set()      # id() of the parameters with a queued (still unwritten) gradient


def _may_defer(weight) -> bool:
    """A deferred gradient is handed to autograd UNWRITTEN: a cast (weight not f32) or a sum with a second gradient of the same
    parameter (one weight feeding two convs) would consume it before the flush -- those run immediately."""
    if weight is None:
        return True
    if weight.dtype != torch.float32 or id(weight) in _deferred_params:
        return False
    _deferred_params.add(id(weight))
    return True


def _defer_wgrad(x, dy, acc, g, key, weight=None, nslabs=0):
    """True if this weight gradient was queued for the batched launch (acc: the zeroed packed accumulator it adds into, or --
    nslabs > 0 -- the slab buffer its pixel splits store into)."""
    if DEFER_WGRAD_MAX_PIX <= 0 or x.shape[0] * x.shape[2] * x.shape[3] > DEFER_WGRAD_MAX_PIX or x.dtype != torch.bfloat16:
        return False
    ok = _wgrad_batchable.get(key)
    if ok is None:
        ok = _wgrad_batchable[key] = bool(lib().npp_conv_wgrad_batchable(_byref(x), _byref(dy), C.byref(g)))
    if not ok or not _may_defer(weight):
        return False
    # the accumulator's MEMORY is kept through an alias with its own TensorImpl: for a 1x1 conv it IS the gradient tensor handed to
    # autograd, and a second reference to that object would make AccumulateGrad clone the (still empty) gradient
    keep = torch.empty(0, dtype=acc.dtype, device=acc.device).set_(acc.untyped_storage(), acc.storage_offset(), acc.shape, acc.stride())
    _pending_wgrads.append((x, dy, keep, g, torch.cuda.current_stream(), int(nslabs)))
    return True


# Slab mode of the batched weight gradients (round 3, OPT-IN: NPP_WGRAD_SLAB_BATCH=1): every pixel split of a job STORES its partial
# tile into a slab of its own and the (batched) unpack sums the slabs: bit-reproducible weight gradients (no float atomics).
# Measured A/B/A/B on one MI355X: 48.9 ms per step against 47.9 with the atomics (weight-gradient tail 8.8 vs 8.0-8.3 ms) -- the
# ~2.3 GB of atomic traffic per step (1.3 TB/s chip-wide) is spread under the MFMA work of the other workgroups, while 2 x 2.3 GB of
# slab stores + reads and the longer unpack are not.  So the accumulate form stays the default.
WGRAD_SLABS_BATCHED = os.environ.get("NPP_WGRAD_SLAB_BATCH", "0") == "1"
_wgrad_batch_splits: dict = {}


def _batched_slabs(x, dy, g, key) -> int:
    """Slabs (= pixel splits) of this problem in the batched launch: in slab mode every problem's, otherwise only those of the
    nine-tap halo kernel (csrc/conv_wgrad_g4.hip: wg9_body), whose splits always store slabs; else 0."""
    if DEFER_WGRAD_MAX_PIX <= 0 or not DEFER_UNPACK or x.dtype != torch.bfloat16:
        return 0
    if x.shape[0] * x.shape[2] * x.shape[3] > DEFER_WGRAD_MAX_PIX:
        return 0
    key = (key, WGRAD_SLABS_BATCHED)
    n = _wgrad_batch_splits.get(key)
    if n is None:
        fn = lib().npp_conv_wgrad_batched_splits if WGRAD_SLABS_BATCHED else lib().npp_conv_wgrad_batched_slabs
        n = _wgrad_batch_splits[key] = int(fn(_byref(x), _byref(dy), C.byref(g)))
    return n


_pending_dw_wgrads: list = []      # depthwise: (x, dy, gradient alias, slab scratch, NppConvGeom, stream)
_dw_batchable: dict = {}
_dw_wgrad_bufs = {"pin": None, "dev": None, "ev": None, "keep": []}


def _defer_dw_wgrad(x, dy, dw, ws, g, weight=None):
    """Depthwise twin of _defer_wgrad (npp_dwconv_bwd_weight_batched: one run launch + one slab-sum launch for all of them)."""
    if DEFER_WGRAD_MAX_PIX <= 0 or x.dtype != torch.bfloat16 or dw.dtype != torch.float32 or not DEFER_DW_WGRAD:
        return False
    key = (tuple(x.shape), L.nhwc_ld(x), tuple(dy.shape), L.nhwc_ld(dy), g.kh, g.sh, g.ph, g.dh, g.relu_in)
    ok = _dw_batchable.get(key)
    if ok is None:
        ok = _dw_batchable[key] = bool(lib().npp_dwconv_bwd_weight_batchable(_byref(x), _byref(dy), C.byref(g)))
    if not ok or not _may_defer(weight):
        return False
    keep = torch.empty(0, dtype=dw.dtype, device=dw.device).set_(dw.untyped_storage(), dw.storage_offset(), dw.shape, dw.stride())
    _pending_dw_wgrads.append((x, dy, keep, ws, g, torch.cuda.current_stream()))
    return True


DEFER_DW_WGRAD = os.environ.get("NPP_DEFER_DW_WGRAD", "1") != "0"


def _table_bufs(B, nb, device, capturing):
    """Pinned + device scratch of a batched launch's job table: (pin, dev), or None when a capture finds none sized by a warm-up step.
    A capture retires the pair (the replayed graph re-reads the pinned image); eager steps wait for the previous upload."""
    if capturing:
        if B["pin"] is None or B["pin"].numel() < nb:
            return None
        pin, dev = B["pin"], B["dev"]
        B["keep"].append((pin, dev))
        B["pin"] = B["dev"] = B["ev"] = None
        return pin, dev
    if B["pin"] is None or B["pin"].numel() < nb:
        if B["ev"] is not None:      # the library's own copy reads the old pinned image: torch's host allocator knows nothing of it
            B["ev"].synchronize()
        pin = torch.empty(max(nb, 1 << 16), dtype=torch.uint8).pin_memory()
        dev = torch.empty(pin.numel(), dtype=torch.uint8, device=device)
        B["pin"], B["dev"], B["ev"] = pin, dev, None
        return pin, dev
    if B["ev"] is not None:
        B["ev"].synchronize()
    return B["pin"], B["dev"]


def _flush_dw_wgrads():
    if not _pending_dw_wgrads:
        return
    items = list(_pending_dw_wgrads)
    _pending_dw_wgrads.clear()
    cur = torch.cuda.current_stream()
    seen = {cur.cuda_stream}
    for it in items:
        if it[5].cuda_stream not in seen:
            seen.add(it[5].cuda_stream)
            cur.wait_stream(it[5])
    n = len(items)
    capturing = torch.cuda.is_current_stream_capturing()
    bufs = _table_bufs(_dw_wgrad_bufs, int(lib().npp_dwconv_bwd_weight_batched_ws(n)), items[0][0].device, capturing)
    if bufs is None:
        for (x, dy, dw, ws, g, _st) in items:
            check(lib().npp_dwconv_bwd_weight(_byref(x), _byref(dy), dw.data_ptr(), ws.data_ptr(), C.byref(g), stream_ptr()),
                  "npp_dwconv_bwd_weight")
        return
    pin, dev = bufs
    arr = (L.NppDwWgradItem * n)()
    for i, (x, dy, dw, ws, g, _st) in enumerate(items):
        arr[i].x, arr[i].dy, arr[i].dw, arr[i].ws, arr[i].g = desc(x), desc(dy), dw.data_ptr(), ws.data_ptr(), g
    check(lib().npp_dwconv_bwd_weight_batched(C.cast(arr, C.c_void_p), n, pin.data_ptr(), dev.data_ptr(), pin.numel(), stream_ptr()),
          "npp_dwconv_bwd_weight_batched")
    if not capturing:
        _dw_wgrad_bufs["ev"] = torch.cuda.Event()
        _dw_wgrad_bufs["ev"].record()
    for it in items:
        if it[5].cuda_stream != cur.cuda_stream:
            for t in it[:4]:
                t.record_stream(cur)


def drop_pending():
    """Forget the deferred launches of a step that was abandoned (a failed hipGraph capture): their tensors are gone."""
    _pending_wgrads.clear()
    _pending_dw_wgrads.clear()
    _pending_se_grads.clear()
    _pending_unpacks.clear()
    _deferred_params.clear()
    for B in (_wgrad_bufs, _wgrad_bufs_k, _dw_wgrad_bufs, _unpack_bufs, _unpack_bufs_k, _se_grad_bufs):
        B["ev"] = None


def flush_wgrads(group=None):
    """Run every deferred weight gradient in one launch per kernel variant on the current stream (before flush_unpacks).
    group "K": only the dense KxK convs (with flush_unpacks after it their gradients are final: GradReducer(overlap="tail") reduces
    them while the rest -- group "O": 1x1, depthwise, SE -- is computed); None: everything."""
    if group != "K":
        _deferred_params.clear()
        _flush_dw_wgrads()
        _flush_se_grads()
    if not _pending_wgrads:
        return
    if group == "K":
        items = [it for it in _pending_wgrads if it[3].kh * it[3].kw > 1]
        rest = [it for it in _pending_wgrads if it[3].kh * it[3].kw <= 1]
        _pending_wgrads.clear()
        _pending_wgrads.extend(rest)
        if not items:
            return
    else:
        items = list(_pending_wgrads)
        _pending_wgrads.clear()
    cur = torch.cuda.current_stream()
    seen = {cur.cuda_stream}
    for it in items:
        if it[4].cuda_stream not in seen:
            seen.add(it[4].cuda_stream)
            cur.wait_stream(it[4])
    n = len(items)
    arr = (L.NppWgradItem * n)()
    for i, (x, dy, acc, g, _st, nsl) in enumerate(items):
        arr[i].x, arr[i].dy, arr[i].dw_packed, arr[i].g, arr[i].nslabs = desc(x), desc(dy), acc.data_ptr(), g, nsl
    nb = int(lib().npp_conv_wgrad_batched_ws(n))
    capturing = torch.cuda.is_current_stream_capturing()
    B = _wgrad_bufs_k if group == "K" else _wgrad_bufs
    if capturing:
        if B["pin"] is None or B["pin"].numel() < nb:      # no warm-up step sized the tables: one launch each
            for (x, dy, acc, g, _st, nsl) in items:
                if nsl > 0:      # (slab buffers are not zeroed: the accumulate form adds into the first slab, the others must be zero)
                    acc.zero_()
                check(lib().npp_conv_wgrad(_byref(x), _byref(dy), acc.data_ptr(), C.byref(g), stream_ptr()), "npp_conv_wgrad")
            return
        pin, dev = B["pin"], B["dev"]
        B["keep"].append((pin, dev))            # the replayed graph re-reads this pinned image: retire it
        B["pin"] = B["dev"] = B["ev"] = None
    elif B["pin"] is None or B["pin"].numel() < nb:
        if B["ev"] is not None:
            B["ev"].synchronize()
        pin = torch.empty(max(nb, 1 << 16), dtype=torch.uint8).pin_memory()
        dev = torch.empty(pin.numel(), dtype=torch.uint8, device=items[0][0].device)
        B["pin"], B["dev"], B["ev"] = pin, dev, None
    else:
        pin, dev = B["pin"], B["dev"]
        if B["ev"] is not None:
            B["ev"].synchronize()               # the previous step's upload has read the pinned image
    check(lib().npp_conv_wgrad_batched(C.cast(arr, C.c_void_p), n, pin.data_ptr(), dev.data_ptr(), pin.numel(), stream_ptr()),
          "npp_conv_wgrad_batched")
    if not capturing:
        B["ev"] = torch.cuda.Event()
        B["ev"].record()
    for it in items:
        if it[4].cuda_stream != cur.cuda_stream:
            it[0].record_stream(cur)
            it[1].record_stream(cur)
            it[2].record_stream(cur)


_unpack_blocks_cache: dict = {}


def _unpack_blocks(co, ci, taps):
    k = (co, ci, taps)
    v = _unpack_blocks_cache.get(k)
    if v is None:
        v = _unpack_blocks_cache[k] = int(lib().npp_unpack_job_blocks(co, ci, taps))
    return v


_unpack_bufs = {"pin": None, "dev": None, "ev": None, "keep": []}
_unpack_bufs_k = {"pin": None, "dev": None, "ev": None, "keep": []}


def flush_unpacks(group=None):
    """Run every deferred unpack in one launch on the current stream (which first waits for the streams they were queued on).
    group "K": only those of KxK weights (a 1x1 weight whose packed rows are padded has an unpack too: its weight gradient belongs
    to the second group of the tail, flush_wgrads)."""
    global _UNPACK_JOB
    if group != "K":
        _deferred_params.clear()
    if not _pending_unpacks:
        return
    import numpy as np
    if _UNPACK_JOB is None:
        _UNPACK_JOB = np.dtype([("src", "<u8"), ("dst", "<u8"), ("cout", "<i4"), ("cin", "<i4"), ("taps", "<i4"), ("cp", "<i4"),
                                ("kpad", "<i4"), ("nslabs", "<i4"), ("slab", "<i8"), ("first_block", "<i8")])
        assert _UNPACK_JOB.itemsize == 56
    if group == "K":
        items = [it for it in _pending_unpacks if it[4] > 1]
        rest = [it for it in _pending_unpacks if it[4] <= 1]
        _pending_unpacks.clear()
        _pending_unpacks.extend(rest)
        if not items:
            return
    else:
        items = list(_pending_unpacks)
        _pending_unpacks.clear()
    cur = torch.cuda.current_stream()
    seen = {cur.cuda_stream}
    for it in items:
        st = it[9]
        if st.cuda_stream not in seen:
            seen.add(st.cuda_stream)
            cur.wait_stream(st)
    n = len(items)
    jobs = np.zeros(n, dtype=_UNPACK_JOB)
    nblk = np.empty(n, dtype=np.int64)
    for i, (src, dst, co, ci, taps, cp, kpad, nsl, slab, _st) in enumerate(items):
        jobs[i] = (src.data_ptr(), dst.data_ptr(), co, ci, taps, cp, kpad, nsl, slab, 0)
        nblk[i] = _unpack_blocks(co, ci, taps)
    first = np.cumsum(nblk) - nblk
    jobs["first_block"] = first
    total = int(nblk.sum())
    bj = np.repeat(np.arange(n, dtype=np.int32), nblk)
    raw = np.concatenate([np.frombuffer(jobs.tobytes(), dtype=np.uint8), np.frombuffer(bj.tobytes(), dtype=np.uint8)])
    nb = raw.size
    capturing = torch.cuda.is_current_stream_capturing()
    B = _unpack_bufs_k if group == "K" else _unpack_bufs
    if capturing:
        # pinned memory cannot be allocated inside a capture: take the image the eager warm-up steps used (same job count) and
        # retire it -- a replayed graph re-reads this pinned image, so it is never rewritten; later eager steps get a new one
        if B["pin"] is None or B["pin"].numel() < nb:
            for (src, dst, co, ci, taps, cp, kpad, nsl, slab, _st) in items:      # no warm-up happened: one launch each
                check(lib().npp_unpack_wgrad_sum(src.data_ptr(), nsl, co, ci, 1, taps, dst.data_ptr(), stream_ptr()),
                      "npp_unpack_wgrad_sum")
            return
        pin, dev = B["pin"], B["dev"]
        B["keep"].append((pin, dev))
        B["pin"] = B["dev"] = B["ev"] = None
    elif B["pin"] is None or B["pin"].numel() < nb:
        if B["ev"] is not None:
            B["ev"].synchronize()
        pin = torch.empty(max(nb, 1 << 16), dtype=torch.uint8).pin_memory()
        dev = torch.empty(pin.numel(), dtype=torch.uint8, device=items[0][1].device)
        B["pin"], B["dev"], B["ev"] = pin, dev, None
    else:
        pin, dev = B["pin"], B["dev"]
        if B["ev"] is not None:
            B["ev"].synchronize()             # the previous step's upload has read the pinned image
    pin.numpy()[:nb] = raw
    dev[:nb].copy_(pin[:nb], non_blocking=True)
    if not capturing:
        B["ev"] = torch.cuda.Event()
        B["ev"].record()
    joff = n * _UNPACK_JOB.itemsize
    check(lib().npp_unpack_wgrad_batched(dev.data_ptr(), dev.data_ptr() + joff, total, stream_ptr()), "npp_unpack_wgrad_batched")
    for it in items:                          # sources / destinations were allocated on the queuing streams
        if it[9].cuda_stream != cur.cuda_stream:
            it[0].record_stream(cur)
            it[1].record_stream(cur)
_wgrad_splits = {}      # shape key -> slabs wanted by the deterministic split-K weight-gradient kernel (0: not taken)


def _conv_launch(x, wp, bf, mask, y, stats, g, s, what, soft=False):
    """npp_conv_fwd with the split-K scratch the library asks for on small feature maps (queried once per shape).
    soft: return NPP_E_UNSUPPORTED instead of raising (a bit-mask on a shape whose kernel cannot read one)."""
    key = (tuple(x.shape), L.nhwc_ld(x), tuple(y.shape), L.nhwc_ld(y), x.dtype, g.kh, g.kw, g.sh, g.sw,
           g.ph, g.pw, g.dh, g.dw, g.uph, g.upw)
    nbytes = _conv_ws_cache.get(key)
    if nbytes is None:
        nbytes = _conv_ws_cache[key] = int(lib().npp_conv_fwd_ws_bytes(_byref(x), _byref(y), C.byref(g)))
    if nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        rc = lib().npp_conv_fwd_ws(_byref(x), wp, bf, mask, _byref(y), stats, C.byref(g), ws.data_ptr(), nbytes, s)
    else:
        rc = lib().npp_conv_fwd(_byref(x), wp, bf, mask, _byref(y), stats, C.byref(g), s)
    if soft and rc == L.NPP_E_UNSUPPORTED:
        return rc
    check(rc, what)
    return 0


SHAPE_LOG = None     # tools/shape_prof.py: list of (kind, n, ci, h, w, co, kh, stride, dil) in launch order


class _Conv2d(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil, relu_in, want_stats, out_hw, fan=None, bias_dead=False):
        x = _gemm_ready(to_nhwc(x))
        n, ci, h, w = x.shape
        co, _, kh, kw = weight.shape
        if out_hw is None:
            oh, ow = _conv_out(h, kh, stride[0], pad[0], dil[0]), _conv_out(w, kw, stride[1], pad[1], dil[1])
        else:
            oh, ow = out_hw
        y = new_nhwc(n, co, oh, ow, x.dtype, x.device)
        stats = stats_buffer(R * 2 * co, x.device, want_stats) if want_stats else None
        g = geom(kh, kw, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], 1, relu_in)
        wp = packed_weight(weight, False, x.dtype)
        bf = None
        if bias is not None:
            bf = bias.detach()
            if bf.dtype != torch.float32:
                bf = bf.float()
        _conv_launch(x, wp.data_ptr(), ptr(bf), None, y, ptr(stats), g, stream_ptr(), "npp_conv_fwd")
        if SHAPE_LOG is not None:
            SHAPE_LOG.append(("fwd", n, ci, h, w, co, kh, kw, stride[0], dil[0]))
        ctx.save_for_backward(x, weight, bias)
        ctx.fan = fan if x.dtype == torch.bfloat16 else None
        ctx.mask_bits = relu_mask_of(x) if (relu_in and RELU_BITS and x.dtype == torch.bfloat16) else None
        if relu_in and ctx.mask_bits is None and RELU_BITS and x.dtype == torch.bfloat16:
            MASK_STATS[2] += 1
        ctx.cfg = (stride, pad, dil, relu_in, bias is not None)
        ctx.bias_dead = bool(bias_dead)
        ctx.set_materialize_grads(False)     # no zero tensor for the (non-differentiable) statistics output
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        if dy is None:
            return (None,) * 11
        x, weight, bias = ctx.saved_tensors
        stride, pad, dil, relu_in, has_bias = ctx.cfg
        dy = to_nhwc(dy)
        if dy.dtype != x.dtype:
            dy = cast(dy, x.dtype)
        dy = _gemm_ready(dy)
        co, _, kh, kw = weight.shape
        dx = dw = db = None
        s = stream_ptr()
        if ctx.needs_input_grad[0]:
            dx = _conv_dgrad(x, dy, packed_weight(weight, True, x.dtype), co, kh, kw, stride, pad, dil, relu_in, ctx.mask_bits, ctx.fan)
        if ctx.needs_input_grad[1]:
            dw = _conv_wgrad(x, dy, weight, stride, pad, dil, relu_in)
        if has_bias and ctx.needs_input_grad[2]:
            db = _bias_grad(dy, bias, co, ctx.bias_dead)
        return dx, dw, db, None, None, None, None, None, None, None, None


# A conv bias directly in front of a BatchNorm that normalises with batch statistics (the 1024 -> 512 / 384 layers, the heads' first
# conv, Pooled_Conv: models/model_augment.py:332-351, 365-398, operations.py:222-251) has the EXACT gradient zero: the BatchNorm
# subtracts the batch mean, which contains the bias, and its backward pass hands down a dy whose sum over the batch vanishes.  The
# reference computes that sum and stores its rounding residue (|.| < 1e-5 of the other gradients in the goldens); here the gradient
# is the exact zero -- no pass over the 96 x 96 x 512 gradient tensors for it.  NPP_BIAS_RESIDUE=1 computes the residue.
BIAS_RESIDUE = os.environ.get("NPP_BIAS_RESIDUE", "0") == "1"


def _bias_grad(dy, bias, co, dead):
    if dead and not BIAS_RESIDUE:
        slot = grad_out(bias, zero=True)      # (a reducer's bucket slot, zeroed by begin_step)
        if slot is not None:
            return slot
        return zeros_f32(co, dy.device, own=True) if bias.dtype == torch.float32 else torch.zeros(co, dtype=bias.dtype, device=dy.device)
    s = stream_ptr()
    acc = zeros_f64(R * co, dy.device)
    check(lib().npp_channel_sum(_byref(dy), acc.data_ptr(), s), "npp_channel_sum")
    db = _grad_buf(bias, co, dy.device)
    check(lib().npp_sum_replicas(acc.data_ptr(), R, co, db.data_ptr(), s), "npp_sum_replicas")
    return db


def _conv_dgrad(x, dy, wp, co, kh, kw, stride, pad, dil, relu_in, mask_bits, fan):
    """Data gradient of y = conv(relu?(x)): the same conv over dy with the flipped-tap image `wp` (co = channels of dy), masked by
    x > 0 (bit-mask when the producer of x left one), stored -- or added into the shared gradient buffer of x's fan-out node."""
    n, ci, h, w = x.shape
    s = stream_ptr()
    dx = None
    acc_flag = 0
    if fan is not None and stride == (1, 1):
        # x feeds several consumers: add into their shared gradient buffer (the first claimant stores)
        dx, accumulate = fan.claim(x)
        acc_flag = 2 if accumulate else 0
    tk = None
    if dx is None:
        dx = new_nhwc(n, ci, h, w, x.dtype, x.device)
    elif BN_SUMS and relu_in and mask_bits is not None and x.dtype == torch.bfloat16:
        tk = fan.last_writer_ticket()
    done = False
    if tk is not None:
        # this launch finishes the gradient of x = BN_a(ya) [+ BN_b(yb)]: its epilogue also delivers the BatchNorm-backward sums
        g = geom(kh, kw, 1, 1, dil[0] * (kh - 1) - pad[0], dil[1] * (kw - 1) - pad[1], dil[0], dil[1],
                 (stride[0], stride[1]), acc_flag)
        sums = zeros_f64(R * (3 if tk.two else 2) * tk.c, x.device)
        sa_ = L.NppBnSumsArgs()
        sa_.ya, sa_.mi_a, sa_.sums, sa_.two = desc(tk.ya), tk.mia.data_ptr(), sums.data_ptr(), int(tk.two)
        if tk.two:
            sa_.yb, sa_.mi_b = desc(tk.yb), tk.mib.data_ptr()
        rc = lib().npp_conv_dgrad_sums(_byref(dy), wp.data_ptr(), C.byref(mask_bits[0]), _byref(dx), C.byref(g), C.byref(sa_), s)
        if rc == 0:
            tk.deliver(sums, dx)
            BN_SUMS_STATS[1] += 1
            MASK_STATS[0] += 1
            done = True
        elif rc != L.NPP_E_UNSUPPORTED:
            check(rc, "npp_conv_dgrad_sums")
    while not done:
        g = geom(kh, kw, 1, 1, dil[0] * (kh - 1) - pad[0], dil[1] * (kw - 1) - pad[1], dil[0], dil[1],
                 (stride[0], stride[1]), acc_flag)
        if relu_in and mask_bits is not None:      # the producer of x left a bit-mask: 1/16 of the mask bytes
            done = _conv_launch(dy, wp.data_ptr(), None, C.byref(mask_bits[0]), dx, None, g, s, "npp_conv_fwd(dgrad)",
                                soft=True) == 0
            if not acc_flag or done:
                MASK_STATS[0 if done else 1] += 1
        if not done:
            done = _conv_launch(dy, wp.data_ptr(), None, _byref(x) if relu_in else None, dx, None, g, s,
                                "npp_conv_fwd(dgrad)", soft=bool(acc_flag)) == 0
        if not done:      # this shape's kernel cannot accumulate: a tensor of its own, the fan-out node adds it
            FAN_STATS[1] -= 1
            FAN_STATS[2] += 1
            fan.broken = True
            acc_flag = 0
            dx = new_nhwc(n, ci, h, w, x.dtype, x.device)
    if SHAPE_LOG is not None:
        SHAPE_LOG.append(("dgrad", n, ci, h, w, co, kh, kw, stride[0], dil[0]))
    if FAN_CENSUS is not None:
        dx._npp_prod = "conv_dgrad"
    return dx


def _conv_wgrad(x, dy, weight, stride, pad, dil, relu_in):
    """Weight gradient of y = conv(relu?(x)) w.r.t. `weight` (dy may be a channel slice of a wider buffer): immediate, or queued for
    the batched launch at the end of backward (TrainStep); returns the tensor autograd receives."""
    n, ci, h, w = x.shape
    co, _, kh, kw = weight.shape
    s = stream_ptr()
    nel = lib().npp_packed_weight_elems(co, ci, kh, kw, 0)
    g = geom(kh, kw, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], 1, relu_in)
    if kh == 1 and kw == 1 and ci % 64 == 0 and co % 32 == 0:
        wkey = (tuple(x.shape), L.nhwc_ld(x), co, L.nhwc_ld(dy), kh, kw, stride, pad, dil, x.dtype)
        nsl = _batched_slabs(x, dy, g, wkey)
        if nsl > 0 and weight.dtype == torch.float32 and _may_defer(weight):
            # batched launch in slab mode: the splits store into slabs, the batched unpack sums them into the gradient
            slabs = torch.empty(nsl * nel, dtype=torch.float32, device=x.device)
            if not _defer_wgrad(x, dy, slabs, g, wkey, None, nsl):
                slabs.zero_()
                check(lib().npp_conv_wgrad(_byref(x), _byref(dy), slabs.data_ptr(), C.byref(g), s), "npp_conv_wgrad")
            dw = grad_out(weight)
            if dw is None:
                dw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
            _unpack_or_defer(slabs, dw, co, ci, 1, 1, nsl, s, True)
        else:
            # packed [co][ci] == OIHW: accumulate straight into the gradient tensor
            dw = grad_out(weight, zero=True)            # the parameter's (zeroed) slot in its gradient bucket, or
            if dw is None:
                dw = zeros_f32(weight.numel(), x.device, own=True).view(weight.shape)     # pre-zeroed pool slice: no fill launch
            if not _defer_wgrad(x, dy, dw, g, wkey, weight):
                check(lib().npp_conv_wgrad(_byref(x), _byref(dy), dw.data_ptr(), C.byref(g), s), "npp_conv_wgrad")
    else:
        dw = grad_out(weight)
        if dw is None:
            dw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
        wkey = (tuple(x.shape), L.nhwc_ld(x), co, L.nhwc_ld(dy), kh, kw, stride, pad, dil, x.dtype)
        nsl = _wgrad_splits.get(wkey)
        if nsl is None:
            nsl = _wgrad_splits[wkey] = int(lib().npp_conv_wgrad_splits(_byref(x), _byref(dy), C.byref(g)))
        defer_ok = DEFER_UNPACK and _may_defer(weight)      # (dw reaches autograd unwritten when the unpack waits)
        bsl = _batched_slabs(x, dy, g, wkey) if (defer_ok and nsl <= 0) else 0
        if bsl > 0:      # batched launch with slabs (the nine-tap halo kernel's jobs always; every job in slab mode)
            slabs = torch.empty(bsl * nel, dtype=torch.float32, device=x.device)
            if not _defer_wgrad(x, dy, slabs, g, wkey, None, bsl):      # (refused after all: the accumulate form into the first slab)
                slabs.zero_()
                check(lib().npp_conv_wgrad(_byref(x), _byref(dy), slabs.data_ptr(), C.byref(g), s), "npp_conv_wgrad")
            _unpack_or_defer(slabs, dw, co, ci, kh, kw, bsl, s, True)
        elif nsl > 0:      # deterministic split-K: the kernel stores one slab per split, the unpack sums them
            slabs = torch.empty(nsl * nel, dtype=torch.float32, device=x.device)
            check(lib().npp_conv_wgrad_slabs(_byref(x), _byref(dy), slabs.data_ptr(), nsl, C.byref(g), s), "npp_conv_wgrad_slabs")
            _unpack_or_defer(slabs, dw, co, ci, kh, kw, nsl, s, defer_ok)
        else:
            dwp = zeros_f32(nel, x.device)
            if not (defer_ok and _defer_wgrad(x, dy, dwp, g, wkey)):      # (its unpack must be deferred behind it)
                check(lib().npp_conv_wgrad(_byref(x), _byref(dy), dwp.data_ptr(), C.byref(g), s), "npp_conv_wgrad")
            _unpack_or_defer(dwp, dw, co, ci, kh, kw, 0, s, defer_ok)
    if SHAPE_LOG is not None:
        SHAPE_LOG.append(("wgrad", n, ci, h, w, co, kh, kw, stride[0], dil[0]))
    if dw.dtype != weight.dtype:
        dw = dw.to(weight.dtype)
    return dw


def conv2d(x, weight, bias=None, stride=1, pad=0, dil=1, relu_in=False, want_stats=False, private_in=False, bias_dead=False):
    """y = conv(relu?(x)) + bias, plus (optionally) the f64 [sum | sumsq] statistics of y.  private_in: `x` has no other
    consumer (an intermediate of the calling module), so it needs no fan-out node.  bias_dead: y goes straight into a BatchNorm that
    uses batch statistics -- the bias gradient is exactly zero (see _bias_grad)."""
    xa, fan = (x, None) if private_in else take_acc(x)
    return _Conv2d.apply(xa, weight, bias, _pair(stride), _pair(pad), _pair(dil), bool(relu_in), int(want_stats), None, fan,
                         bool(bias_dead))


def conv2d_crop(x, weight, stride=2, relu_in=False, want_stats=False):
    """1x1 strided conv of the view x[:, :, 1:, 1:] (FactorizedReduce.conv2, operations.py:155) without
    materialising the view: the crop is a padding of -1 in the gather."""
    h, w = x.shape[2] - 1, x.shape[3] - 1
    out_hw = ((h - 1) // stride + 1, (w - 1) // stride + 1)
    return _Conv2d.apply(take(x), weight, None, _pair(stride), (-1, -1), (1, 1), bool(relu_in), int(want_stats), out_hw)


# --------------------------------------------------------------------------------------------------
# merged edges: the ReLU-conv-BN edges of a cell that read the SAME state with the SAME geometry as ONE conv launch
# --------------------------------------------------------------------------------------------------
# The fixed genotype hands this over on a plate (genotypes.py:30-54): three of the eight edges of an encoder cell are std_conv_3x3 on
# state 0, the decoder cells apply three (two) std_conv_1x1 to state 1 (0), the refinement cells two std_conv_3x3 to the same state
# twice over.  Forward: ONE conv C -> m C whose weight image is the edges' own images back to back (rows = output channels), with one
# statistics epilogue over m C channels; every edge's raw output is a channel slice of the one result.  Backward: the BatchNorm
# backward of each edge writes its dy straight into its slice of ONE buffer (`gslot`), the data gradient is ONE conv m C -> C with
# K = m * taps * C -- the sum over the edges happens in the MFMA accumulators, no read-add-store pass -- and the weight gradients
# stay per parameter (dy slices with ld = m C).  The parameters remain separate tensors under the reference's names; only the
# derived operand images are shared.  NPP_WIDE=0 keeps one launch per edge.
WIDE = os.environ.get("NPP_WIDE", "1") != "0"
WIDE_SYNC = os.environ.get("NPP_WIDE_SYNC", "1") != "0"      # merged edges under SyncBatchNorm too (their statistics rows travel as one segment)
WIDE_STATS = [0, 0, 0]      # merged forward launches / merged data gradients that found every dy in place / ... that had to gather


class _WideGrad:
    """The dy buffer [N, H, W, sum of the members' Cout] of one merged conv: allocated when the first edge's BatchNorm backward asks
    for its slice."""
    __slots__ = ("buf", "cos", "offs", "stream", "member_stat", "bias_dead")

    def __init__(self, cos, bias_dead=False):
        self.buf, self.cos, self.stream, self.member_stat = None, list(cos), 0, None
        self.bias_dead = bool(bias_dead)      # every member feeds a BatchNorm on batch statistics: bias gradients are exactly zero
        self.offs = [sum(self.cos[:k]) for k in range(len(self.cos))]

    def slot(self, k):
        wg = self

        def give(like):
            n, c, h, w = like.shape
            if c != wg.cos[k]:
                return None
            if wg.buf is None:
                wg.buf = new_nhwc(n, sum(wg.cos), h, w, like.dtype, like.device)
                wg.stream = stream_ptr() if like.is_cuda else 0
            elif like.is_cuda and stream_ptr() != wg.stream:
                wg.buf.record_stream(torch.cuda.current_stream())      # (an edge whose BatchNorm backward runs on the other branch's stream)
            b = wg.buf
            if b.shape[0] != n or b.shape[2] != h or b.shape[3] != w or b.dtype != like.dtype or b.device != like.device:
                return None
            return _alias(b, wg.offs[k], c)
        return give


def _dx_into(slot, like):
    """The tensor a BatchNorm backward writes the gradient of its raw input into: the edge's slice of a merged conv's dy buffer when
    there is one, else a fresh tensor."""
    if slot is not None:
        t = slot(like)
        if t is not None:
            return t
    return new_nhwc(*like.shape, like.dtype, like.device)


class WideGroup:
    """m conv modules (same Cin, kernel, padding, stride 1) that are applied to one and the same tensor.
    Forward: consecutive members without bias and of equal Cout form a RUN that is one conv Cin -> (run width) (the weights' forward
    images back to back); every other member -- another width, a bias: the 1024 -> 512 and 1024 -> 384 layers on the concatenated
    decoder features, model_augment.py:332-351; the head conv that reads what two refinement cells read -- keeps a launch of its
    own (tiles chosen per width).  separate_fwd=True forces that for all members.
    Backward: ONE data gradient for the whole group, sum(Cout) -> Cin over the concatenated dy (the sum over the consumers happens
    in the MFMA accumulators instead of read-add-store passes over the gradient tensor)."""

    def __init__(self, convs, separate_fwd=False):
        self.convs = list(convs)
        self.separate_fwd = bool(separate_fwd)
        self.img = {}          # ("d", dtype) / ("f", run index, dtype) -> image written by the model's WeightPacker (valid after its pack_if_stale)
        self.calls = None      # per-forward cache: [input tensor, list of pending results, stream, event]
        runs = []
        for k, c in enumerate(self.convs):
            co = c.weight.shape[0]
            if (not self.separate_fwd and runs and c.bias is None and self.convs[runs[-1][-1]].bias is None
                    and self.convs[runs[-1][-1]].weight.shape[0] == co and co % 32 == 0):
                runs[-1].append(k)
            else:
                runs.append([k])
        self.runs = runs

    @property
    def weights(self):
        return [c.weight for c in self.convs]

    @property
    def cos(self):
        return [c.weight.shape[0] for c in self.convs]

    def images(self, dtype, need_dgrad):
        """([forward image | None per run], data-gradient image | None) of the merged conv."""
        ws = self.weights
        dev = ws[0].device
        merged = [r for r, run in enumerate(self.runs) if len(run) > 1]
        dg = self.img.get(("d", dtype))
        if dg is not None and dg.device == dev and all(("f", r, dtype) in self.img for r in merged):
            return [self.img.get(("f", r, dtype)) for r in range(len(self.runs))], dg
        # no packer manages this group (a cell used on its own): build the images from the per-weight ones
        _, ci, kh, kw = ws[0].shape
        taps = kh * kw
        fwd = []
        for run in self.runs:
            if len(run) == 1:
                fwd.append(None)
                continue
            co = ws[run[0]].shape[0]
            n1 = co * ((taps * ((ci + 7) // 8 * 8) + 63) // 64 * 64)
            fwd.append(torch.cat([packed_weight(ws[k], False, dtype).view(-1)[:n1] for k in run]))
        dg = None
        if need_dgrad:
            rows = (ci + 31) // 32 * 32
            ct = sum(self.cos)
            ctp = (ct + 7) // 8 * 8
            kg = (taps * ctp + 63) // 64 * 64
            dg = torch.zeros(rows, kg, dtype=dtype, device=dev)
            v = dg[:, :taps * ctp].view(rows, taps, ctp)
            off = 0
            for w in ws:
                co = w.shape[0]
                cop1 = (co + 7) // 8 * 8
                k1 = (taps * cop1 + 63) // 64 * 64
                one = packed_weight(w, True, dtype).view(rows, k1)[:, :taps * cop1].view(rows, taps, cop1)[:, :, :co]
                v[:, :, off:off + co] = one
                off += co
            dg = dg.view(-1)
        return fwd, dg


class _ConvWide(Function):
    """Args: x, pad, relu_in, want_stats, fan, group, wg, then the m weights, then the m biases (None where a conv has none)."""

    @staticmethod
    def forward(ctx, x, pad, relu_in, want_stats, fan, group, wg, *wb):
        x = _gemm_ready(to_nhwc(x))
        n, ci, h, w = x.shape
        m = len(wb) // 2
        weights, biases = wb[:m], wb[m:]
        cos = [wt.shape[0] for wt in weights]
        _, _, kh, kw = weights[0].shape
        g = geom(kh, kw, 1, 1, pad[0], pad[1], 1, 1, 1, relu_in)
        need_bwd = any(ctx.needs_input_grad)      # (grad mode is off inside forward: ask the context)
        wps, wpd = group.images(x.dtype, need_bwd)
        s = stream_ptr()
        outs, stats_all = [None] * m, []
        member_stat = [None] * m                  # per member: (index into stats_all, offset, statistics row width | 0)
        for r, run in enumerate(group.runs):
            if len(run) == 1:
                k = run[0]
                wt = weights[k]
                yk = new_nhwc(n, cos[k], h, w, x.dtype, x.device)
                st = stats_buffer(R * 2 * cos[k], x.device, want_stats) if want_stats else None
                bf = None
                if biases[k] is not None:
                    bf = biases[k].detach()
                    if bf.dtype != torch.float32:
                        bf = bf.float()
                _conv_launch(x, packed_weight(wt, False, x.dtype).data_ptr(), ptr(bf), None, yk, ptr(st), g, s, "npp_conv_fwd")
                if SHAPE_LOG is not None:
                    SHAPE_LOG.append(("fwd", n, ci, h, w, cos[k], kh, kw, 1, 1))
                outs[k] = yk
                member_stat[k] = (len(stats_all), 0, 0)
                stats_all.append(st)
            else:
                co, mr = cos[run[0]], len(run)
                y = new_nhwc(n, mr * co, h, w, x.dtype, x.device)
                st = stats_buffer(R * 2 * mr * co, x.device, want_stats) if want_stats else None
                _conv_launch(x, wps[r].data_ptr(), None, None, y, ptr(st), g, s, "npp_conv_fwd(wide)")
                if SHAPE_LOG is not None:
                    SHAPE_LOG.append(("fwd", n, ci, h, w, mr * co, kh, kw, 1, 1))
                for q, k in enumerate(run):
                    outs[k] = _alias(y, q * co, co)
                    member_stat[k] = (len(stats_all), q * co, mr * co)
                stats_all.append(st)
        wg.member_stat = member_stat
        outs = tuple(outs)
        WIDE_STATS[0] += 1
        ctx.save_for_backward(x, *weights, *[b for b in biases if b is not None])
        ctx.has_bias = [b is not None for b in biases]
        ctx.wpd = wpd
        ctx.fan = fan if x.dtype == torch.bfloat16 else None
        ctx.mask_bits = relu_mask_of(x) if (relu_in and RELU_BITS and x.dtype == torch.bfloat16) else None
        if relu_in and ctx.mask_bits is None and RELU_BITS and x.dtype == torch.bfloat16:
            MASK_STATS[2] += 1
        ctx.cfg = (pad, relu_in, m, cos)
        ctx.wg = wg
        ctx.set_materialize_grads(False)
        live = [st for st in stats_all if st is not None]
        if live:
            ctx.mark_non_differentiable(*live)
        return (*outs, *stats_all)

    @staticmethod
    def backward(ctx, *grads):
        pad, relu_in, m, cos = ctx.cfg
        saved = ctx.saved_tensors
        x, weights = saved[0], saved[1:1 + m]
        bl = list(saved[1 + m:])
        biases = [bl.pop(0) if hb else None for hb in ctx.has_bias]
        dys = list(grads[:m])
        if all(d is None for d in dys):
            return (None,) * (7 + 2 * m)
        n, ci, h, w = x.shape
        _, _, kh, kw = weights[0].shape
        wg = ctx.wg
        buf, wg.buf = wg.buf, None      # (a second backward over the same graph starts a new buffer)
        esz = x.element_size()
        offs = wg.offs
        in_place = buf is not None and buf.dtype == x.dtype and all(
            d is not None and d.dtype == buf.dtype and d.shape[1] == cos[k] and d.data_ptr() == buf.data_ptr() + offs[k] * esz
            and d.stride() == (buf.stride(0), 1, buf.stride(2), buf.stride(3)) for k, d in enumerate(dys))
        ct = sum(cos)
        if in_place:
            dy_all = buf
            WIDE_STATS[1] += 1
        else:      # (an edge whose BatchNorm ran on a path without slots, or an unused edge): gather the slices
            parts = []
            for k, d in enumerate(dys):
                if d is None:
                    parts.append(new_nhwc(n, cos[k], h, w, x.dtype, x.device, zero=True))
                else:
                    d = to_nhwc(d)
                    parts.append(d if d.dtype == x.dtype else cast(d, x.dtype))
            dy_all = new_nhwc(n, ct, h, w, x.dtype, x.device)
            descs = [desc(t) for t in parts]
            arr = (C.POINTER(L.NppTensor) * m)(*[C.pointer(d) for d in descs])
            check(lib().npp_concat(arr, m, _byref(dy_all), stream_ptr()), "npp_concat")
            WIDE_STATS[2] += 1
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _conv_dgrad(x, dy_all, ctx.wpd, ct, kh, kw, (1, 1), pad, (1, 1), relu_in, ctx.mask_bits, ctx.fan)
        dws, dbs = [], []
        s = stream_ptr()
        merged = None
        if kh == 1 and kw == 1 and all(d is not None for d in dys) and all(ctx.needs_input_grad[7 + k] for k in range(m)):
            merged = _conv_wgrad_merged(x, dy_all, weights, cos, relu_in)
        for k, wt in enumerate(weights):
            dyk = _alias(dy_all, offs[k], cos[k]) if dys[k] is not None else None
            if merged is not None:
                dws.append(merged[k])
            elif ctx.needs_input_grad[7 + k] and dyk is not None:
                dws.append(_conv_wgrad(x, dyk, wt, (1, 1), pad, (1, 1), relu_in))
            else:
                dws.append(None)
            if biases[k] is not None and ctx.needs_input_grad[7 + m + k] and dyk is not None:
                dbs.append(_bias_grad(dyk, biases[k], cos[k], wg.bias_dead))
            else:
                dbs.append(None)
        return (dx, None, None, None, None, None, None, *dws, *dbs)


# The 1x1 members of a merged edge read the same x: as separate jobs of the batched weight-gradient launch each of them fetches it from
# HBM for itself (1024 channels @96^2 = 302 MB, twice per branch for the layer pair; 151 MB three times for a refinement pair + head).
# Their dy are channel slices of ONE buffer, so the group is ONE weight-gradient problem with Cout = sum of the members': the output
# tiles of a pixel split run side by side on one XCD and share the x rows through its L2.  The job accumulates into a packed
# [sum Cout][Cin] scratch, the batched unpack copies each member's rows into its gradient (a few MB).  NPP_WIDE_WGRAD=0: one job each.
WIDE_WGRAD = os.environ.get("NPP_WIDE_WGRAD", "1") != "0"
WIDE_WGRADS = [0]      # merged weight-gradient jobs queued


def _conv_wgrad_merged(x, dy_all, weights, cos, relu_in):
    """One deferred weight-gradient job for the 1x1 members of a merged edge: the members' gradient tensors, or None (not applicable:
    the caller queues one job per member)."""
    n, ci, h, w = x.shape
    ct = sum(cos)
    if not (WIDE_WGRAD and DEFER_UNPACK and 0 < n * h * w <= DEFER_WGRAD_MAX_PIX and x.dtype == torch.bfloat16 and ci % 64 == 0):
        return None
    if any(wt.dtype != torch.float32 or id(wt) in _deferred_params for wt in weights):
        return None
    g = geom(1, 1, 1, 1, 0, 0, 1, 1, 1, relu_in)
    wkey = (tuple(x.shape), L.nhwc_ld(x), ct, L.nhwc_ld(dy_all), 1, 1, (1, 1), (0, 0), (1, 1), x.dtype, "merged")
    ok = _wgrad_batchable.get(wkey)
    if ok is None:
        ok = _wgrad_batchable[wkey] = bool(lib().npp_conv_wgrad_batchable(_byref(x), _byref(dy_all), C.byref(g)))
    if not ok:
        return None
    for wt in weights:
        _may_defer(wt)
    kpad = (ci + 63) // 64 * 64
    dwp = zeros_f32(ct * kpad, x.device)
    keep = torch.empty(0, dtype=dwp.dtype, device=dwp.device).set_(dwp.untyped_storage(), dwp.storage_offset(), dwp.shape, dwp.stride())
    _pending_wgrads.append((x, dy_all, keep, g, torch.cuda.current_stream(), 0))
    s = stream_ptr()
    outs, off = [], 0
    for wt, co in zip(weights, cos):
        dw = grad_out(wt)
        if dw is None:
            dw = torch.empty(wt.shape, dtype=torch.float32, device=x.device)
        _unpack_or_defer(dwp[off * kpad:(off + co) * kpad], dw, co, ci, 1, 1, 0, s, True)
        outs.append(dw)
        off += co
    WIDE_WGRADS[0] += 1
    return outs


def conv2d_wide(x, group, pad, relu_in, want_stats, bias_dead=False):
    """The m convs of `group` applied to x (one launch per run of the group, see WideGroup) with ONE merged data gradient:
    ([raw output k], [statistics (view) k], [statistics row width k | 0], [dy slot k])."""
    xa, fan = take_acc(x)
    ws = group.weights
    bs = [c.bias for c in group.convs]
    m, cos = len(ws), group.cos
    wg = _WideGrad(cos, bias_dead)
    res = _ConvWide.apply(xa, _pair(pad), bool(relu_in), int(want_stats), fan, group, wg, *ws, *bs)
    ys, stats_list = res[:m], res[m:]
    svs, scs = [], []
    for k in range(m):
        idx, off, width = wg.member_stat[k]
        st = stats_list[idx]
        svs.append(st[off:] if st is not None else None)
        scs.append(width if st is not None else 0)
    return ys, svs, scs, [wg.slot(k) for k in range(m)]


# --------------------------------------------------------------------------------------------------
# depthwise conv
# --------------------------------------------------------------------------------------------------
class _DwConv2d(Function):
    @staticmethod
    def forward(ctx, x, weight, stride, pad, dil, relu_in, fan=None):
        x = to_nhwc(x)
        ctx.fan = fan if x.dtype == torch.bfloat16 else None
        n, c, h, w = x.shape
        _, _, kh, kw = weight.shape
        oh, ow = _conv_out(h, kh, stride, pad, dil), _conv_out(w, kw, stride, pad, dil)
        y = new_nhwc(n, c, oh, ow, x.dtype, x.device)
        g = geom(kh, kw, stride, stride, pad, pad, dil, dil, 1, relu_in)
        wf = weight.detach()
        if wf.dtype != torch.float32 or not wf.is_contiguous():
            wf = wf.float().contiguous()
        check(lib().npp_dwconv_fwd(_byref(x), wf.data_ptr(), _byref(y), C.byref(g), stream_ptr()), "npp_dwconv_fwd")
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad, dil, relu_in)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, pad, dil, relu_in = ctx.cfg
        dy = to_nhwc(dy)
        if dy.dtype != x.dtype:
            dy = cast(dy, x.dtype)
        _, _, kh, kw = weight.shape
        g = geom(kh, kw, stride, stride, pad, pad, dil, dil, 1, relu_in)
        s = stream_ptr()
        wf = weight.detach()
        if wf.dtype != torch.float32 or not wf.is_contiguous():
            wf = wf.float().contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx, accumulate = _claim_dx(ctx.fan if stride == 1 else None, x)
            if accumulate:
                ga = geom(kh, kw, stride, stride, pad, pad, dil, dil, 1, int(relu_in) | 2)
                rc = lib().npp_dwconv_bwd_data(_byref(dy), wf.data_ptr(), _byref(x) if relu_in else None, _byref(dx), C.byref(ga), s)
                if rc == L.NPP_E_UNSUPPORTED:
                    _unclaim()
                    ctx.fan.broken = True
                    dx, accumulate = new_nhwc(*x.shape, x.dtype, x.device), False
                else:
                    check(rc, "npp_dwconv_bwd_data")
            if not accumulate:
                check(lib().npp_dwconv_bwd_data(_byref(dy), wf.data_ptr(), _byref(x) if relu_in else None, _byref(dx),
                                                C.byref(g), s), "npp_dwconv_bwd_data")
        if ctx.needs_input_grad[1]:
            dw = grad_out(weight)
            if dw is None:
                dw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
            nws = lib().npp_dwconv_bwd_weight_ws(_byref(dy), C.byref(g))
            if lib().npp_dwconv_bwd_weight_ws_zeroed(_byref(dy), C.byref(g)):
                ws = zeros_f32(nws, x.device)
            else:
                ws = torch.empty(nws, dtype=torch.float32, device=x.device)
            if not _defer_dw_wgrad(x, dy, dw, ws, g, weight):
                check(lib().npp_dwconv_bwd_weight(_byref(x), _byref(dy), dw.data_ptr(), ws.data_ptr(), C.byref(g), s),
                      "npp_dwconv_bwd_weight")
            if dw.dtype != weight.dtype:
                dw = dw.to(weight.dtype)
        return dx, dw, None, None, None, None, None


def dwconv2d(x, weight, stride=1, pad=0, dil=1, relu_in=False):
    xa, fan = take_acc(x)
    return _DwConv2d.apply(xa, weight, int(stride), int(pad), int(dil), bool(relu_in), fan)


# --------------------------------------------------------------------------------------------------
# batch-norm apply fused with the branch add:  out = relu?( A(a) + B(b) )
# --------------------------------------------------------------------------------------------------
class BnSide:
    """One operand of the fused add.  kind 'bn': `x` is a raw (pre-BN) tensor with f64 stats (train) or a
    BatchNorm holder in eval mode; kind 'plain': `x` is used as is."""

    __slots__ = ("x", "bn", "stats", "count", "synced_ws", "private", "stream", "sync_event", "stats_c", "gslot", "rider", "ticket", "mates", "carry", "home")

    def __init__(self, x, bn=None, stats=None, private=None, stats_c=0, gslot=None, rider=False):
        self.x = x
        self.bn = bn
        # merged edges (conv2d_wide): `stats` starts at this edge's first channel inside statistics rows of stats_c channels
        # ([R][sum stats_c | sumsq stats_c]); gslot(like) -> the edge's slice of the merged conv's dy buffer (BatchNorm backward)
        self.stats_c = int(stats_c)
        self.gslot = gslot
        self.ticket = None          # (out) the _BnTicket of the fused add this side went into, see BN_SUMS
        # SyncBatchNorm + merged edges: the statistics rows of ALL the merged edges travel as the segment of the run's first edge;
        # the others ride along (they wait in the pool to learn that the exchange happened, but contribute no segment of their own)
        self.rider = bool(rider)
        self.mates = None           # merged edges: every side of this side's run (statistics row), itself included
        self.carry = None           # a rider promoted to carrier (its lead edge's exchange was folded into a kernel): the run's full rows
        self.home = None            # raw handle of the stream whose sync pool holds the un-exchanged statistics
        # private: `x` was produced for this operand alone (a raw conv / pool output on its way into its BatchNorm), so it
        # needs no fan-out node (take): ~600 autograd nodes per step less on the host
        self.private = (bn is not None) if private is None else bool(private)
        self.stats = stats
        self.count = None
        self.synced_ws = 0
        self.stream = None          # (shared sync pool) the stream whose kernels produced the statistics
        self.sync_event = None      # (shared sync pool) completion of the exchange that carried them
        if stats is not None and bn is not None and _sync_pool.cur().holds_unsynced(stats):
            grp, ws = _sync_group(bn)
            if grp is not None:
                _sync_pool.cur().enlist(self, grp, ws)
                if x.is_cuda:
                    self.home = torch.cuda.current_stream().cuda_stream


_SYNC_EVEN_ALONE = False   # test hook: run the SyncBN collectives on a 1-rank group (bench.py --force-dist)
SYNC_OFF = False           # ablation (bench.py `exposed_comm_ms`, NPP_LOCAL_BN of SURVEY §8e): SyncBatchNorm modules use LOCAL statistics


def _sync_group(bn):
    """SyncBatchNorm (augment_lip_sync.py:191) -> the process group to reduce statistics over."""
    if SYNC_OFF:
        return None, 1
    if isinstance(bn, torch.nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized():
        grp = bn.process_group
        ws = dist.get_world_size(grp) if grp is not None else dist.get_world_size()
        if ws > 1 or _SYNC_EVEN_ALONE:
            return grp if grp is not None else dist.group.WORLD, ws
    return None, 1


def stats_level(bn) -> int:
    """0: running statistics; 1: batch statistics; 2: batch statistics exchanged across ranks."""
    if not (bn.training or bn.running_mean is None):
        return 0
    return 2 if _sync_group(bn)[0] is not None else 1


# SyncBatchNorm exchanges INSIDE the fused kernels (csrc/p2p_xp.h): with the statistics going through the peer-to-peer mailboxes the
# fused apply / backward-apply kernels trade their local sums for the world's in their own prologue -- the stand-alone exchange launch
# in front of them (446 links of the step's dependent chain) disappears.  NPP_P2P_FOLD=0: every exchange a launch of its own.
P2P_FOLD = os.environ.get("NPP_P2P_FOLD", "1") != "0"
FOLD_STATS = [0, 0]      # exchanges folded into a forward / a backward kernel


def _fold_channel(n_doubles, grp):
    """The mailbox channel for an in-kernel exchange of n_doubles on the current stream, or -1."""
    if not (P2P_FOLD and P2P_DIRECT and FUSE_BN_SYNC and not SYNC_MERGE):
        return -1
    from . import comm
    return comm.p2p_fold_channel(n_doubles, grp)


def _leave_pool(sd, pools=None):
    """Take `sd` off the waiting list of the sync pool that holds its statistics (it is exchanged on its own: inside a fused kernel, or
    alone).  (The other edges of a merged conv that still wait do not depend on it: with P2P_FOLD a flush sends every such edge as a
    private copy of its own slice, see _SyncStatsPool._flush_slabs.)"""
    for pl in (_sync_pool.all() if pools is None else pools):
        if any(w is sd for w in pl.waiting):
            pl.waiting = [w for w in pl.waiting if w is not sd]
            return pl
    return None


def fold_ready(grp) -> bool:
    """The fused BatchNorm kernels can carry the exchanges of SyncBatchNorms of `grp` themselves (operations.WideEdges: merged edges
    whose members are picked up on two streams are only possible then -- each consumer exchanges its own slice on its own channel)."""
    if not (P2P_FOLD and P2P_DIRECT and FUSE_BN_SYNC and FUSE_BN_FIN and not SYNC_MERGE):
        return False
    from . import comm
    return comm.p2p_active() and grp is comm._p2p["group"]


def _fold_forward(launch_sides, training: bool):
    """launch_sides: the (sa, sb) pairs of ONE fused apply launch.  If every BatchNorm side is a SyncBatchNorm of one group whose
    statistics are still local (written by the producing kernel into NPP_STAT_REPLICAS slabs) and the mailboxes take the vector, the
    sides are marked exchanged (world count, off the pool's waiting list) and the channel is returned: the launch MUST then carry the
    exchange.  -1: nothing changed, the caller exchanges as before (_presync_stats)."""
    if not (P2P_FOLD and P2P_DIRECT and FUSE_BN_SYNC and training and not SYNC_MERGE):
        return -1
    ch = _fold_forward_try(launch_sides)
    if ch < 0 and _FOLD_DBG is not None:
        _FOLD_DBG[_fold_why(launch_sides)] += 1
    return ch


_FOLD_DBG = __import__("collections").Counter() if os.environ.get("NPP_DBG_FOLD") else None
if _FOLD_DBG is not None:
    import atexit as _atexit
    _atexit.register(lambda: [print(f"npp-fold-refused {n:5d} {k}", file=__import__("sys").stderr) for k, n in _FOLD_DBG.most_common()])


def _fold_why(launch_sides):
    out = []
    for sa, sb in launch_sides:
        for sd in (sa, sb):
            if sd is None or sd.bn is None:
                out.append("plain")
            else:
                st = sd.stats
                out.append(f"sync={_sync_group(sd.bn)[0] is not None} synced={sd.synced_ws} ev={sd.sync_event is not None} "
                           f"stats={'none' if st is None else st.numel()} c={sd.x.shape[1]} sc={sd.stats_c}")
    return " | ".join(out)


def _fold_forward_try(launch_sides):
    grp = None
    bn_sides, widths = [], set()
    for sa, sb in launch_sides:
        ns = 0
        for sd in (sa, sb):
            if sd is None or sd.bn is None:
                continue
            g_, ws = _sync_group(sd.bn)
            if g_ is None or (grp is not None and g_ is not grp) or sd.synced_ws or sd.sync_event is not None or sd.stats is None:
                return -1
            st = sd.stats
            c = sd.x.shape[1]
            if not st.is_cuda or st.dtype != torch.float64 or (not sd.stats_c and st.numel() != R * 2 * c):
                return -1
            grp = g_
            bn_sides.append((sd, ws))
            ns += 1
        widths.add((ns, sa.x.shape[1]))
    if not bn_sides or len(widths) != 1:
        return -1
    ns, c = next(iter(widths))
    ch = _fold_channel(len(launch_sides) * ns * 2 * c, grp)
    if ch < 0:
        return -1
    pools = _sync_pool.all()
    for sd, ws in bn_sides:
        _leave_pool(sd, pools)
        sd.synced_ws = ws
    return ch


def _presync_stats(sides, training: bool):
    """SyncBN forward exchange.  Statistics written by the producing kernels into the sync pool are all-reduced in place,
    replicas and all, together with every other vector produced since the previous exchange (one collective per
    fused-add node or fewer, zero glue kernels); anything else (FactorizedReduce's stitched halves, statistics computed
    on demand) is all-reduced on its own."""
    for sd in sides:
        if sd is not None and sd.synced_ws and sd.sync_event is not None:
            # exchanged by a merged collective (possibly triggered by the other branch): this stream must see it complete
            if sd.x.is_cuda:
                torch.cuda.current_stream().wait_event(sd.sync_event)
            sd.sync_event = None
        if sd is None or sd.bn is None or sd.synced_ws or not (training or sd.bn.running_mean is None):
            continue
        grp, ws = _sync_group(sd.bn)
        if grp is None:
            continue
        if sd.stats is None:
            sd.stats = channel_stats(sd.x, 2)
            if _sync_pool.cur().holds_unsynced(sd.stats):
                _sync_pool.cur().enlist(sd, grp, ws)
        owner = next((pl for pl in _sync_pool.all() if any(w is sd for w in pl.waiting)), None)
        if owner is not None and sd.home is not None and sd.x.is_cuda and sd.home != torch.cuda.current_stream().cuda_stream:
            # statistics that sit in ANOTHER stream's pool (an edge of a merged conv picked up on the other task branch's stream; this
            # stream has waited for the conv): never flush that pool from here -- its other vectors belong to kernels this stream
            # has not waited for.  This side alone: a copy of its rows, exchanged on this stream
            _leave_pool(sd, [owner])
            if sd.stats_c:
                _compact_stats(sd)
            else:
                sd.stats = sd.stats.clone()
            hub_all_reduce(sd.stats, grp)
            sd.synced_ws = ws
        elif owner is not None:
            owner.flush()
        else:
            hub_all_reduce(sd.stats, grp)
            sd.synced_ws = ws


def _bn_finalize_args(side: BnSide, training: bool, device):
    """Everything npp_bn_finalize needs for one side that normalises with batch statistics (None otherwise): the
    argument struct plus the tensors it points to (kept alive by the caller)."""
    bn = side.bn
    if not (training or bn.running_mean is None):
        return None
    c = side.x.shape[1]
    if not side.synced_ws:
        _presync_stats((side,), training)     # SyncBatchNorm: all-reduce (computing the statistics if need be)
    if side.stats is None:
        side.stats = channel_stats(side.x)
    stats = side.stats
    if side.stats_c:      # a slice of a merged conv's statistics rows: R replicas of 2 * stats_c doubles, this edge's channels at `stats`
        nrep = R
    else:
        nrep = stats.numel() // (2 * c)
    count = float(side.x.shape[0] * side.x.shape[2] * side.x.shape[3])
    if side.synced_ws:            # all-reduced: the replicas now hold global partial sums
        count *= side.synced_ws
    side.count = count
    ss = torch.empty(2 * c, dtype=torch.float32, device=device)
    mi = torch.empty(2 * c, dtype=torch.float32, device=device)
    gamma = bn.weight.detach() if bn.weight is not None else None
    beta = bn.bias.detach() if bn.bias is not None else None
    track = bn.track_running_stats and bn.running_mean is not None and training
    mom = bn.momentum if bn.momentum is not None else 0.1
    nbt = bn.num_batches_tracked if (track and bn.num_batches_tracked is not None) else None
    args = L.NppBnFinalizeArgs(stats.data_ptr(), ptr(gamma), ptr(beta), ptr(bn.running_mean) if track else None,
                               ptr(bn.running_var) if track else None, ptr(nbt), ss.data_ptr(), mi.data_ptr(), count, nrep,
                               float(mom), float(bn.eps), side.stats_c)
    return args, ss, mi, (stats, gamma, beta)


def _compact_stats(side: BnSide):
    """A merged edge's statistics as a [R][2C] tensor of its own (the separate finalize kernels read rows of 2C doubles)."""
    if side.stats_c and side.stats is not None:
        c, sc = side.x.shape[1], side.stats_c
        rows = side.stats.as_strided((R, 2, c), (2 * sc, sc, 1))
        side.stats = rows.contiguous().view(-1)
        side.stats_c = 0


def _bn_coeffs(side: BnSide, training: bool, device):
    """scale/shift (+ mean/invstd) for one BN side; updates running stats in train mode."""
    bn = side.bn
    c = side.x.shape[1]
    _compact_stats(side)
    prep = _bn_finalize_args(side, training, device)
    if prep is not None:
        a, ss, mi, _keep = prep
        check(lib().npp_bn_finalize(a.stats, a.nrep, a.count, a.gamma, a.beta, a.running_mean, a.running_var,
                                    a.num_batches_tracked, a.momentum, a.eps, a.scale_shift, a.mean_invstd, c, stream_ptr()),
              "npp_bn_finalize")
        return ss, mi, True
    ss = torch.empty(2 * c, dtype=torch.float32, device=device)
    gamma = bn.weight.detach() if bn.weight is not None else None
    beta = bn.bias.detach() if bn.bias is not None else None
    check(lib().npp_bn_eval_coeffs(ptr(gamma), ptr(beta), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                   float(bn.eps), ss.data_ptr(), c, stream_ptr()), "npp_bn_eval_coeffs")
    mi = None
    if torch.is_grad_enabled():   # mean / invstd for the (rare) backward in eval mode
        mi = torch.cat([bn.running_mean.detach().float(), torch.rsqrt(bn.running_var.detach().float() + bn.eps)])
    return ss, mi, False


def _bn_coeffs_pair(sa: BnSide, sb: BnSide, training: bool, device):
    """Both sides of a two-sided add: one npp_bn_finalize2 launch when both use batch statistics."""
    use_a = training or sa.bn.running_mean is None
    use_b = training or sb.bn.running_mean is None
    if use_a and use_b and sa.x.shape[1] == sb.x.shape[1]:
        _compact_stats(sa)
        _compact_stats(sb)
        pa = _bn_finalize_args(sa, training, device)
        pb = _bn_finalize_args(sb, training, device)
        check(lib().npp_bn_finalize2(C.byref(pa[0]), C.byref(pb[0]), sa.x.shape[1], stream_ptr()), "npp_bn_finalize2")
        return (pa[1], pa[2], True), (pb[1], pb[2], True)
    return _bn_coeffs(sa, training, device), _bn_coeffs(sb, training, device)


FUSE_BN_FIN = os.environ.get("NPP_FUSE_BN_FIN", "1") != "0"
_fused_ok_cache: dict = {}


def _fused_layout_ok(t) -> bool:
    key = (t.shape[1], L.nhwc_ld(t), t.dtype)
    ok = _fused_ok_cache.get(key)
    if ok is None:
        ok = _fused_ok_cache[key] = bool(lib().npp_bn_fused_ok(_byref(t)))
    return ok


def _local_batch_bn(bn, training: bool) -> bool:
    return bn is not None and (training or bn.running_mean is None) and _sync_group(bn)[0] is None


FUSE_BN_SYNC = os.environ.get("NPP_FUSE_BN_SYNC", "1") != "0"      # SyncBatchNorm too: the finalize / coefficients AFTER the exchange


def _fusable_batch_bn(bn, training: bool) -> bool:
    """Batch statistics, local -- or exchanged (SyncBatchNorm): by the time the consumer runs the replicas hold the global sums and
    the count is the global count, so the same prologue applies."""
    if bn is None or not (training or bn.running_mean is None):
        return False
    return FUSE_BN_SYNC or _sync_group(bn)[0] is None


def _fin_fusable(sa, sb, a, b, training: bool) -> bool:
    """out = BN_a(a) [+ BN_b(b) | + b] with batch statistics on every BatchNorm side, in a layout npp_affine_add_fin takes."""
    if not FUSE_BN_FIN or not _fusable_batch_bn(sa.bn, training) or not _fused_layout_ok(a):
        return False
    if sb is not None:
        if sb.bn is not None and (not _fusable_batch_bn(sb.bn, training) or sb.x.shape[1] != sa.x.shape[1]):
            return False
        if b is None or b.dtype != a.dtype or b.shape != a.shape or not _fused_layout_ok(b):
            return False
    return True


# BatchNorm backward of a small map in one launch (csrc/bn_one.hip): the grid barrier's counters belong to ONE stream (its launches
# are stream-ordered) and are never reset; allocated outside any graph capture on first use (TrainStep warms up eagerly first).
# OPT-IN (NPP_BN_ONE=1).  Measured on one MI355X, 50 dependent two-sided BatchNorm backwards replayed from a hipGraph, N = 16
# (tools/bn_time.py; us per backward, two launches / one launch / one launch with the barrier compiled out):
#     32 ch @96^2  17.7 / 36.1 / 20.8      64 ch @48^2  14.4 / 26.3 / 12.2      128 ch @24^2  15.8 / 20.3 / 10.9      256 ch @12^2  16.9 / 18.9 / 13.3
# and the whole step 47.3 ms against 46.4.  A grid-wide barrier is a chain of device-scope round trips through the fabric that links
# the 8 XCDs (the blocks' f64 atomics must have landed, arrive, poll, read the sums back: ~1.5-2 us each), 6-14 us in all, whether
# the arrivals go to one counter or to a two-level tree -- a kernel boundary costs ~3 us.  The two-launch form stays the default.
BN_ONE = os.environ.get("NPP_BN_ONE", "0") == "1"
_bn_one_ctr: dict = {}
_bn_one_plan: dict = {}
BN_ONE_STATS = [0, 0]      # launches of the one-sided / two-sided kernel (tests)


def _bn_one_blocks(x, two):
    if not BN_ONE or x.dtype != torch.bfloat16:
        return 0
    key = (x.shape[0] * x.shape[2] * x.shape[3], x.shape[1], two)
    nb = _bn_one_plan.get(key)
    if nb is None:
        nb = _bn_one_plan[key] = int(lib().npp_bn_bwd_one_blocks(key[0], key[1], L.NPP_BF16, 1 if two else 0))
    return nb


def _bn_one_barrier(device):
    st = torch.cuda.current_stream()
    key = (device.index, st.cuda_stream)
    buf = _bn_one_ctr.get(key)
    if buf is None:
        if torch.cuda.is_current_stream_capturing():
            return None      # (never allocate the counters from a graph's private pool)
        buf = _bn_one_ctr[key] = torch.zeros(24 * 257, dtype=torch.int64, device=device)
    return buf


def _fin_prepare(a, b, sa, sb, training, out):
    """(finalize args a, finalize args b | None, output tensor) of the fused apply npp_affine_add_fin, or None when the operands are
    not local-batch-statistics BatchNorms in a layout it takes."""
    if not _fin_fusable(sa, sb, a, b, training):
        return None
    dev = a.device
    pa = _bn_finalize_args(sa, training, dev)
    pb = _bn_finalize_args(sb, training, dev) if (sb is not None and sb.bn is not None) else None
    y = out[0] if out is not None else new_nhwc(*a.shape, a.dtype, dev)
    return pa, pb, y


def _fin_record(ctx, a, b, y, pa, pb, sa, sb, relu):
    """What _BnAdd.backward needs after a fused apply."""
    ctx.relu = relu
    ctx.sides = (sa.bn, sb.bn if sb is not None else None, True, pb is not None, sa.count, sb.count if sb else None, b is not None)
    ctx.save_for_backward(a, b, y if relu else None, pa[2], pb[2] if pb is not None else None, None, None)
    # BN_SUMS: the last writer of y's gradient may deliver the backward sums (no ReLU on y, bf16, both operands BatchNorms of one shape
    # or one BatchNorm + nothing / a plain operand -- the forms the fused backward kernels take)
    tk = None
    if BN_SUMS and not relu and a.dtype == torch.bfloat16:
        two = pb is not None
        if (not two) or (b is not None and b.shape == a.shape and b.dtype == a.dtype):
            tk = _BnTicket(a, b if two else None, pa[2], pb[2] if two else None)
            BN_SUMS_STATS[0] += 1
    ctx.ticket = tk
    sa.ticket = tk


# One launch for the BatchNorm applies / backward passes of SEVERAL independent, equally shaped nodes (npp_affine_add_fin_multi,
# npp_bn_bwd_reduce_multi, npp_bn_bwd_apply_multi): the nodes of a cell that are ready together.  NPP_BN_MULTI=0: one launch each.
BN_MULTI = os.environ.get("NPP_BN_MULTI", "1") != "0"
BN_MULTI_MAX = 4
MULTI_STATS = [0, 0, 0, 0]      # multi-job launches: forward applies / backward reduces / backward applies / SyncBatchNorm exchanges between the two
BN_MULTI_SYNC = os.environ.get("NPP_BN_MULTI_SYNC", "1") != "0"      # the multi-job backward under SyncBatchNorm (p2p mailboxes); 0: one reduce + apply per node


def _zero_desc():
    return L.NppTensor(None, 0, 0, 0, 0, 0, 0, 0)


class _BnAdd(Function):
    """out = relu?( [BN_a](a) + [BN_b](b) ).  Tensor args: a, gamma_a, beta_a, b, gamma_b, beta_b."""

    @staticmethod
    def forward(ctx, a, ga, ba, b, gb, bb, sa: BnSide, sb: Optional[BnSide], relu: bool, training: bool, out=None, mk=None):
        # out: None, or a 1-element list holding the tensor to write (a channel slice of a cell's output buffer, see
        # ConcatBuffer); wrapped so that autograd does not see an in-place write into a view
        dev = a.device
        ssa = mia = ssb = mib = None
        batch_a = batch_b = False
        ctx.gslots = (sa.gslot, sb.gslot if sb is not None else None)
        xch = -1
        if _fin_fusable(sa, sb, a, b, training) and (mk is None or a.dtype == torch.bfloat16):
            xch = _fold_forward([(sa, sb)], training)      # SyncBatchNorm: the exchange inside the fused kernel
        if xch < 0:
            _presync_stats((sa, sb), training)
        prep = _fin_prepare(a, b, sa, sb, training, out)
        if prep is not None:
            # local train-mode BatchNorm(s): the finalize arithmetic runs in the prologue of the affine_add kernel
            pa, pb, y = prep
            rc = lib().npp_affine_add_fin_x(_byref(y), _byref(a), C.byref(pa[0]), tref(b), C.byref(pb[0]) if pb is not None else None,
                                            int(relu), (mk[0].data_ptr() + mk[1]) if mk is not None else None,
                                            mk[2] if mk is not None else 0, xch, stream_ptr())
            if rc == 0:
                if xch >= 0:
                    FOLD_STATS[0] += 1
                _fin_record(ctx, a, b, y, pa, pb, sa, sb, relu)
                return y
            if rc != -5 or xch >= 0:      # NPP_E_UNSUPPORTED: nothing was launched, the separate kernels take it (never with a folded exchange)
                check(rc, "npp_affine_add_fin_x")
        elif xch >= 0:
            raise RuntimeError("a SyncBatchNorm exchange was folded into a fused apply that did not take place")
        if sa.bn is not None and sb is not None and sb.bn is not None:
            (ssa, mia, batch_a), (ssb, mib, batch_b) = _bn_coeffs_pair(sa, sb, training, dev)
        else:
            if sa.bn is not None:
                ssa, mia, batch_a = _bn_coeffs(sa, training, dev)
            if sb is not None and sb.bn is not None:
                ssb, mib, batch_b = _bn_coeffs(sb, training, dev)
        y = out[0] if out is not None else new_nhwc(*a.shape, a.dtype, dev)
        if mk is not None:      # (mask bytes, byte offset, bytes per pixel): the kernel also writes y's ReLU bit-mask
            check(lib().npp_affine_add_m(_byref(y), _byref(a), ptr(ssa), tref(b), ptr(ssb), int(relu),
                                         mk[0].data_ptr() + mk[1], mk[2], stream_ptr()), "npp_affine_add_m")
        else:
            check(lib().npp_affine_add(_byref(y), _byref(a), ptr(ssa), tref(b), ptr(ssb), int(relu), stream_ptr()),
                  "npp_affine_add")
        ctx.relu = relu
        ctx.ticket = None
        sa.ticket = None
        ctx.sides = (sa.bn, sb.bn if sb is not None else None, batch_a, batch_b, sa.count, sb.count if sb else None,
                     b is not None)
        ctx.save_for_backward(a, b, y if relu else None, mia, mib, ssa, ssb)
        return y

    @staticmethod
    def backward(ctx, dout):
        gen = _BnAdd._backward_gen(ctx, dout)
        try:
            req = next(gen)
            while True:      # the body asks for a peer-to-peer slab exchange (SyncBatchNorm): do it here, alone
                from . import comm
                req = gen.send(comm.p2p_exchange_slabs(req[0], req[1]))
        except StopIteration as done:
            return done.value

    @staticmethod
    def _backward_gen(ctx, dout):
        """backward() as a generator: it yields (segments, group) where it needs a peer-to-peer slab exchange and is sent whether
        the exchange took place -- _BnAddPair.backward runs two of these side by side and merges their exchanges into one."""
        a, b, yrelu, mia, mib, ssa, ssb = ctx.saved_tensors
        bna, bnb, batch_a, batch_b, cnt_a, cnt_b, has_b = ctx.sides
        gs = getattr(ctx, "gslots", (None, None))      # merged edges: where the raw gradients go (_WideGrad.slot)
        dout = to_nhwc(dout)
        if dout.dtype != a.dtype:
            dout = cast(dout, a.dtype)
        s = stream_ptr()

        def plain_side(x, need_x):
            if not need_x:
                return None
            if yrelu is None:
                # the incoming gradient itself is handed on (possibly to BOTH operands, and kept by deferred weight gradients
                # below them): from here on it has several readers, nobody may add into it in place (_mark_owned)
                try:
                    dout._npp_own = False
                except Exception:      # noqa: BLE001
                    pass
                return dout
            dx = new_nhwc(*x.shape, x.dtype, x.device)
            check(lib().npp_scale_mask(_byref(dout), None, _byref(yrelu), _byref(dx), s), "npp_scale_mask")
            return dx

        def reduce_side(x, mi, acc=False):
            c = x.shape[1]
            nb = lib().npp_reduce_blocks(x.shape[0] * x.shape[2] * x.shape[3], c, L.npp_dtype(x.dtype))
            if acc:      # the blocks add into R zeroed slabs: bn_bwd_apply_fin sums them in its prologue
                sums = zeros_f64(R * 2 * c, x.device)
                check(lib().npp_bn_bwd_reduce_acc(_byref(dout), _byref(x), tref(yrelu), mi.data_ptr(), sums.data_ptr(), nb, s),
                      "npp_bn_bwd_reduce_acc")
                return sums, R
            sums = torch.empty(nb * 2 * c, dtype=torch.float64, device=x.device)   # one slab per block, written
            check(lib().npp_bn_bwd_reduce(_byref(dout), _byref(x), tref(yrelu), mi.data_ptr(), sums.data_ptr(), nb, s),
                  "npp_bn_bwd_reduce")
            return sums, nb

        ni = ctx.needs_input_grad
        # both sides SyncBatchNorm of one group, statistics through the peer-to-peer mailboxes: the two-sided kernels apply as well --
        # the exchange kernel collapses the reduce's slabs, writes the LOCAL dgamma / dbeta and leaves the world's sums in replica 0
        sync2 = None
        if (has_b and bna is not None and bnb is not None and P2P_DIRECT and dout.is_cuda and FUSE_BN_SYNC
                and _sync_group(bna)[0] is not None and _sync_group(bna)[0] is _sync_group(bnb)[0]):
            from . import comm as _comm2
            if _comm2.p2p_can(3 * a.shape[1], _sync_group(bna)[0]):
                sync2 = _sync_group(bna)[0]
        if (has_b and bna is not None and bnb is not None and batch_a and batch_b and ni[0] and ni[3]
                and a.shape == b.shape and a.dtype == b.dtype and cnt_a == cnt_b
                and ((_sync_group(bna)[0] is None and _sync_group(bnb)[0] is None) or sync2 is not None)
                and (sync2 is None or (FUSE_BN_FIN and _fused_layout_ok(a) and _fused_layout_ok(b) and _fused_layout_ok(dout)
                                       and (yrelu is None or _fused_layout_ok(yrelu))))):
            # both edges end in (local) BatchNorm: the two sides share dout and the ReLU mask -> two-sided kernels
            c = a.shape[1]
            dev = a.device
            nb = lib().npp_reduce_blocks(a.shape[0] * a.shape[2] * a.shape[3], c, L.npp_dtype(a.dtype))
            fused = (FUSE_BN_FIN and _fused_layout_ok(a) and _fused_layout_ok(b) and _fused_layout_ok(dout)
                     and (yrelu is None or _fused_layout_ok(yrelu)))
            one = None
            if fused and yrelu is None and sync2 is None and _bn_one_blocks(a, True) > 0:
                one = _bn_one_barrier(dev)
            if one is not None:
                sums = zeros_f64(R * 3 * c, dev)
            elif fused:
                tk_ = getattr(ctx, "ticket", None)
                sums = tk_.take(dout, True) if tk_ is not None else None      # delivered by the gradient's last writer? (LOCAL sums: exchanged below under SyncBatchNorm)
                if sums is None:
                    sums = zeros_f64(R * 3 * c, dev)
                    check(lib().npp_bn_bwd_reduce2_acc(_byref(dout), _byref(a), _byref(b), tref(yrelu), mia.data_ptr(), mib.data_ptr(),
                                                       sums.data_ptr(), nb, s), "npp_bn_bwd_reduce2_acc")
            else:
                sums = torch.empty(nb * 3 * c, dtype=torch.float64, device=dev)
                check(lib().npp_bn_bwd_reduce2(_byref(dout), _byref(a), _byref(b), tref(yrelu), mia.data_ptr(), mib.data_ptr(),
                                               sums.data_ptr(), nb, s), "npp_bn_bwd_reduce2")
            dgb_ = torch.empty(4 * c, dtype=torch.float32, device=dev)
            dgb_ = [dgb_[:c], dgb_[c:2 * c], dgb_[2 * c:3 * c], dgb_[3 * c:]]
            for i_, prm in enumerate((bna.weight, bna.bias, bnb.weight, bnb.bias)):
                slot = grad_out(prm) if ni[(1, 2, 4, 5)[i_]] else None
                if slot is not None:
                    dgb_[i_] = slot
            ga = bna.weight.detach() if bna.weight is not None else None
            gb = bnb.weight.detach() if bnb.weight is not None else None
            if one is not None:
                dxa = _dx_into(gs[0], a)
                dxb = _dx_into(gs[1], b)
                check(lib().npp_bn_bwd_one2(_byref(dout), _byref(a), _byref(b), sums.data_ptr(), float(cnt_a), mia.data_ptr(),
                                            mib.data_ptr(), ptr(ga), ptr(gb), dgb_[0].data_ptr(), dgb_[1].data_ptr(),
                                            dgb_[2].data_ptr(), dgb_[3].data_ptr(), _byref(dxa), _byref(dxb), one.data_ptr(), s),
                      "npp_bn_bwd_one2")
                BN_ONE_STATS[1] += 1
                return (dxa, dgb_[0] if ni[1] else None, dgb_[1] if ni[2] else None,
                        dxb, dgb_[2] if ni[4] else None, dgb_[3] if ni[5] else None, None, None, None, None, None, None)
            if fused and sync2 is not None:
                xch = _fold_channel(3 * c, sync2)
                if xch >= 0:      # the exchange inside the apply's prologue: its leader workgroup writes the LOCAL dgamma / dbeta
                    dxa = _dx_into(gs[0], a)
                    dxb = _dx_into(gs[1], b)
                    check(lib().npp_bn_bwd_apply2_fin_x(_byref(dout), _byref(a), _byref(b), tref(yrelu), sums.data_ptr(), R, float(cnt_a),
                                                        mia.data_ptr(), mib.data_ptr(), ptr(ga), ptr(gb), dgb_[0].data_ptr(),
                                                        dgb_[1].data_ptr(), dgb_[2].data_ptr(), dgb_[3].data_ptr(), _byref(dxa),
                                                        _byref(dxb), xch, s), "npp_bn_bwd_apply2_fin_x")
                    FOLD_STATS[1] += 1
                    return (dxa, dgb_[0] if ni[1] else None, dgb_[1] if ni[2] else None,
                            dxb, dgb_[2] if ni[4] else None, dgb_[3] if ni[5] else None, None, None, None, None, None, None)
                # local sums -> dbeta (both sides), dgamma_a, dgamma_b; world sums -> replica 0, the other replicas zeroed
                ok2 = yield ([(sums, 3 * c, R, c, (dgb_[1], dgb_[3], dgb_[0], dgb_[2]), True)], sync2)
                assert ok2, "peer-to-peer exchange refused a vector it had accepted the size of"
                dxa = _dx_into(gs[0], a)
                dxb = _dx_into(gs[1], b)
                check(lib().npp_bn_bwd_apply2_fin(_byref(dout), _byref(a), _byref(b), tref(yrelu), sums.data_ptr(), R, float(cnt_a),
                                                  mia.data_ptr(), mib.data_ptr(), ptr(ga), ptr(gb), None, None, None, None,
                                                  _byref(dxa), _byref(dxb), s), "npp_bn_bwd_apply2_fin")
                return (dxa, dgb_[0] if ni[1] else None, dgb_[1] if ni[2] else None,
                        dxb, dgb_[2] if ni[4] else None, dgb_[3] if ni[5] else None, None, None, None, None, None, None)
            if fused:
                dxa = _dx_into(gs[0], a)
                dxb = _dx_into(gs[1], b)
                check(lib().npp_bn_bwd_apply2_fin(_byref(dout), _byref(a), _byref(b), tref(yrelu), sums.data_ptr(), R, float(cnt_a),
                                                  mia.data_ptr(), mib.data_ptr(), ptr(ga), ptr(gb), dgb_[0].data_ptr(),
                                                  dgb_[1].data_ptr(), dgb_[2].data_ptr(), dgb_[3].data_ptr(), _byref(dxa),
                                                  _byref(dxb), s), "npp_bn_bwd_apply2_fin")
                return (dxa, dgb_[0] if ni[1] else None, dgb_[1] if ni[2] else None,
                        dxb, dgb_[2] if ni[4] else None, dgb_[3] if ni[5] else None, None, None, None, None, None, None)
            co = torch.empty(6 * c, dtype=torch.float32, device=dev)
            check(lib().npp_bn_bwd_coeffs2(sums.data_ptr(), nb, float(cnt_a), mia.data_ptr(), mib.data_ptr(), ptr(ga), ptr(gb),
                                           co[:3 * c].data_ptr(), co[3 * c:].data_ptr(), dgb_[0].data_ptr(),
                                           dgb_[1].data_ptr(), dgb_[2].data_ptr(), dgb_[3].data_ptr(), c, s),
                  "npp_bn_bwd_coeffs2")
            dxa = _dx_into(gs[0], a)
            dxb = _dx_into(gs[1], b)
            check(lib().npp_bn_bwd_apply2(_byref(dout), _byref(a), _byref(b), tref(yrelu), co[:3 * c].data_ptr(),
                                          co[3 * c:].data_ptr(), _byref(dxa), _byref(dxb), s), "npp_bn_bwd_apply2")
            return (dxa, dgb_[0] if ni[1] else None, dgb_[1] if ni[2] else None,
                    dxb, dgb_[2] if ni[4] else None, dgb_[3] if ni[5] else None, None, None, None, None, None, None)
        sides = [(a, bna, mia, ssa, batch_a, cnt_a, ni[0], ni[1], ni[2])]
        if has_b:
            sides.append((b, bnb, mib, ssb, batch_b, cnt_b, ni[3], ni[4], ni[5]))
        # phase 1: local reductions of every BN side
        lay_ok = FUSE_BN_FIN and _fused_layout_ok(dout) and (yrelu is None or _fused_layout_ok(yrelu))
        fin = [bn is not None and batch and lay_ok and need_x and _sync_group(bn)[0] is None and _fused_layout_ok(x)
               for (x, bn, mi, ss, batch, count, need_x, *_r) in sides]
        # SyncBatchNorm: slabs -> npp_bn_bwd_sum -> all-reduce as before, then the coefficients in the apply's prologue (one vector)
        fin_sync = [bn is not None and batch and lay_ok and need_x and FUSE_BN_SYNC and _sync_group(bn)[0] is not None
                    and _fused_layout_ok(x) for (x, bn, mi, ss, batch, count, need_x, *_r) in sides]
        # a small map with a local BatchNorm and no ReLU mask: reduce + apply in one launch (csrc/bn_one.hip)
        one_bar = [None] * len(sides)
        if yrelu is None:
            for i, (x, bn, mi, ss, batch, count, need_x, *_r) in enumerate(sides):
                if fin[i] and _bn_one_blocks(x, False) > 0:
                    one_bar[i] = _bn_one_barrier(x.device)
        # SyncBatchNorm sides whose sums will travel through the peer-to-peer mailboxes: ACC reduce (R slabs), the exchange kernel
        # collapses the slabs itself (comm.p2p_exchange_slabs) -- no npp_bn_bwd_sum launch
        p2p_slabs = False
        if P2P_DIRECT and dout.is_cuda and any(fin_sync):
            from . import comm as _comm
            p2p_slabs = _comm.p2p_can(4 * a.shape[1], _sync_group(bna if bna is not None else bnb)[0]) and all(fs or sides[i][1] is None or not sides[i][4] or _sync_group(sides[i][1])[0] is None
                                                   for i, fs in enumerate(fin_sync))
        tk_ = getattr(ctx, "ticket", None)
        got = None
        if (tk_ is not None and not tk_.two and len(sides) >= 1 and sides[0][1] is not None and (fin[0] or (p2p_slabs and fin_sync[0])) and one_bar[0] is None
                and not (has_b and sides[-1][1] is not None and len(sides) > 1)):
            got = tk_.take(dout, False)      # side a's sums were delivered by the gradient's last writer
        red = [((got, R) if (i == 0 and got is not None) else
                (zeros_f64(R * 2 * x.shape[1], x.device), R) if one_bar[i] is not None
                else reduce_side(x, mi, fin[i] or (p2p_slabs and fin_sync[i])))
               if bn is not None else None for i, (x, bn, mi, *_r) in enumerate(sides)]
        # phase 2 (SyncBatchNorm): collapse each side's slabs to one vector (+ the LOCAL dgamma / dbeta, which DDP
        # averages afterwards as torch.nn.SyncBatchNorm does), then ONE all-reduce of the side(s) of this node.
        local = [None] * len(sides)
        sync = []
        for i, (x, bn, mi, ss, batch, count, *_r) in enumerate(sides):
            if bn is None or not batch:
                continue
            grp, ws = _sync_group(bn)
            if grp is not None:
                sync.append((i, grp, x.shape[1]))
        folded = [False] * len(sides)
        if sync and p2p_slabs and P2P_FOLD and not SYNC_MERGE and len({id(g_) for _, g_, _ in sync}) == 1:
            # every side's exchange inside its own apply launch (phase 3): nothing travels here
            for i, grp, c in sync:
                bn_i = sides[i][1]
                need_g_, need_b_ = sides[i][7], sides[i][8]
                local[i] = (_grad_buf(bn_i.weight if need_g_ else None, c, dout.device),
                            _grad_buf(bn_i.bias if need_b_ else None, c, dout.device))
                folded[i] = grp
            sync = []
        if sync and p2p_slabs:
            segs = []
            for i, grp, c in sync:
                bn_i = sides[i][1]
                need_g_, need_b_ = sides[i][7], sides[i][8]
                dgl = (_grad_buf(bn_i.weight if need_g_ else None, c, dout.device),
                       _grad_buf(bn_i.bias if need_b_ else None, c, dout.device))
                segs.append((red[i][0], 2 * c, red[i][1], c, (dgl[1], None, dgl[0], None), False))      # [0, c): sum dout -> dbeta; [c, 2c): -> dgamma
                local[i] = dgl
            one_group = len({id(g_) for _, g_, _ in sync}) == 1
            took = (yield (segs, sync[0][1])) if one_group else False
            if took:
                for i, grp, c in sync:
                    red[i] = (red[i][0], 1)
                sync = []
            else:      # (mailboxes too small / two groups: collapse and exchange separately, below)
                local = [None] * len(sides)
        if sync:
            tot = torch.empty(sum(2 * c for _, _, c in sync), dtype=torch.float64, device=dout.device)
            off = 0
            for i, grp, c in sync:
                bn_i = sides[i][1]
                need_g_, need_b_ = sides[i][7], sides[i][8]
                dgl = (_grad_buf(bn_i.weight if need_g_ else None, c, dout.device),
                       _grad_buf(bn_i.bias if need_b_ else None, c, dout.device))
                part = tot[off:off + 2 * c]
                check(lib().npp_bn_bwd_sum(red[i][0].data_ptr(), red[i][1], part.data_ptr(), dgl[0].data_ptr(),
                                           dgl[1].data_ptr(), c, s), "npp_bn_bwd_sum")
                local[i] = dgl
                red[i] = (part, 1)
                off += 2 * c
            if len(sync) == 2 and sync[0][1] is not sync[1][1]:
                for i, grp, c in sync:
                    hub_all_reduce(red[i][0], grp)
            else:
                hub_all_reduce(tot, sync[0][1])
        # phase 3: coefficients + apply
        outs = []
        for i, (x, bn, mi, ss, batch, count, need_x, need_g, need_b) in enumerate(sides):
            if bn is None:
                outs.append((plain_side(x, need_x), None, None))
                continue
            c = x.shape[1]
            sums, nrep = red[i]
            gamma = bn.weight.detach() if bn.weight is not None else None
            dx = dg = db = None
            if batch:
                dgt = dbt = None
                if local[i] is not None:
                    dg, db = local[i]
                else:
                    dgt = _grad_buf(bn.weight if need_g else None, c, x.device)
                    dbt = _grad_buf(bn.bias if need_b else None, c, x.device)
                    dg, db = dgt, dbt
                if one_bar[i] is not None:
                    dx = _dx_into(gs[i], x)
                    check(lib().npp_bn_bwd_one(_byref(dout), _byref(x), sums.data_ptr(), float(count), mi.data_ptr(), ptr(gamma),
                                               ptr(dgt), ptr(dbt), _byref(dx), one_bar[i].data_ptr(), s), "npp_bn_bwd_one")
                    BN_ONE_STATS[0] += 1
                    outs.append((dx, dg if need_g else None, db if need_b else None))
                    continue
                if folded[i]:
                    xch = _fold_channel(2 * c, folded[i])
                    if xch < 0:
                        raise RuntimeError("the mailboxes refused an in-kernel exchange they had accepted the size of")
                    dx = _dx_into(gs[i], x)
                    check(lib().npp_bn_bwd_apply_fin_x(_byref(dout), _byref(x), tref(yrelu), sums.data_ptr(), nrep, float(count),
                                                       mi.data_ptr(), ptr(gamma), dg.data_ptr(), db.data_ptr(), _byref(dx), xch, s),
                          "npp_bn_bwd_apply_fin_x")
                    FOLD_STATS[1] += 1
                    outs.append((dx, dg if need_g else None, db if need_b else None))
                    continue
                if fin[i] or (fin_sync[i] and nrep == 1):
                    dx = _dx_into(gs[i], x)
                    check(lib().npp_bn_bwd_apply_fin(_byref(dout), _byref(x), tref(yrelu), sums.data_ptr(), nrep, float(count),
                                                     mi.data_ptr(), ptr(gamma), ptr(dgt), ptr(dbt), _byref(dx), s),
                          "npp_bn_bwd_apply_fin")
                    outs.append((dx, dg if need_g else None, db if need_b else None))
                    continue
                co = torch.empty(3 * c, dtype=torch.float32, device=x.device)
                check(lib().npp_bn_bwd_coeffs(sums.data_ptr(), nrep, float(count), mi.data_ptr(), ptr(gamma), co.data_ptr(),
                                              ptr(dgt), ptr(dbt), c, s), "npp_bn_bwd_coeffs")
                if need_x:
                    dx = _dx_into(gs[i], x)
                    check(lib().npp_bn_bwd_apply(_byref(dout), _byref(x), tref(yrelu), co.data_ptr(), _byref(dx), s),
                          "npp_bn_bwd_apply")
            else:
                if need_x:
                    dx = _dx_into(gs[i], x)
                    check(lib().npp_scale_mask(_byref(dout), ss.data_ptr(), tref(yrelu), _byref(dx), s), "npp_scale_mask")
                tot = sums.view(nrep, 2 * c).sum(0)
                dg, db = tot[c:].float(), tot[:c].float()
            outs.append((dx, dg if need_g else None, db if need_b else None))
        da, dga, dba = outs[0]
        dbx = dgb = dbb = None
        if has_b:
            dbx, dgb, dbb = outs[1]
        return da, dga, dba, dbx, dgb, dbb, None, None, None, None, None, None


class _SubCtx:
    """What _BnAdd.forward / _backward_gen need of an autograd context, for one half of a _BnAddPair."""

    def __init__(self, saved=None, relu=False, sides=None, needs=None, gslots=(None, None), ticket=None):
        self.saved_tensors = saved
        self.relu, self.sides, self.needs_input_grad = relu, sides, needs
        self.gslots = gslots
        self.ticket = ticket

    def save_for_backward(self, *ts):
        self.saved_tensors = ts


class _BnAddPair(Function):
    """n independent fused BatchNorm adds (the two preprocess outputs of a cell, two nodes of one wave, the candidates of a mixed
    edge) as ONE autograd node, so that under SyncBatchNorm their backward passes share one statistics exchange: every reduce runs,
    ONE peer-to-peer slab exchange carries all the sums (up to 8 segments per launch), the applies follow.
    Args: n, then (a, gamma_a, beta_a, b, gamma_b, beta_b) x n, then the n (sa, sb, relu, training, out, mk) tuples of _BnAdd.forward."""

    @staticmethod
    def forward(ctx, n, *args):
        tens, metas = args[:6 * n], args[6 * n:7 * n]
        subs, outs, saved = [], [], []
        ctx.n = n
        if BN_MULTI and 2 <= n <= BN_MULTI_MAX and _BnAddPair._forward_multi(ctx, n, tens, metas, outs):
            return tuple(outs)
        outs = []
        for k in range(n):
            sub = _SubCtx()
            outs.append(_BnAdd.forward(sub, *tens[6 * k:6 * k + 6], *metas[k]))
            subs.append((sub.relu, sub.sides, sub.gslots, sub.ticket))
            saved += list(sub.saved_tensors)
        assert len(saved) == 7 * n
        ctx.save_for_backward(*saved)
        ctx.subs, ctx.n = subs, n
        return tuple(outs)

    @staticmethod
    def _forward_multi(ctx, n, tens, metas, outs) -> bool:
        """All n applies in ONE launch; False (nothing launched) when one of them is not a fused local apply or the shapes differ."""
        preps = []
        xch = -1
        pat = {(tuple(tens[6 * k].shape), tens[6 * k].dtype, tens[6 * k + 3] is None,
                metas[k][1] is not None and metas[k][1].bn is not None, metas[k][5] is None, metas[k][3]) for k in range(n)}
        if len(pat) == 1 and all(_fin_fusable(metas[k][0], metas[k][1], tens[6 * k], tens[6 * k + 3], metas[k][3]) for k in range(n)) \
                and (next(iter(pat))[4] or tens[0].dtype == torch.bfloat16):
            # one shape, one operand pattern, every apply fusable: under SyncBatchNorm the ONE launch carries the exchange of all jobs
            xch = _fold_forward([(metas[k][0], metas[k][1]) for k in range(n)], metas[0][3])
        for k in range(n):
            a, _ga, _ba, b, _gb, _bb = tens[6 * k:6 * k + 6]
            sa, sb, relu, training, out, mk = metas[k]
            if xch < 0:
                _presync_stats((sa, sb), training)
            pr = _fin_prepare(a, b, sa, sb, training, out)
            if pr is None:
                if xch >= 0:
                    raise RuntimeError("a SyncBatchNorm exchange was folded into a multi-job apply that did not take place")
                return False
            preps.append(pr)
        jobs = (L.NppAffineAddJob * n)()
        for k in range(n):
            a, _ga, _ba, b, _gb, _bb = tens[6 * k:6 * k + 6]
            _sa, _sb, relu, _training, _out, mk = metas[k]
            pa, pb, y = preps[k]
            j = jobs[k]
            j.out, j.a = desc(y), desc(a)
            j.b = desc(b) if b is not None else _zero_desc()
            j.fin_a = pa[0]
            if pb is not None:
                j.fin_b = pb[0]
            j.relu = int(relu)
            if mk is not None:
                j.mask_bits, j.ld_mask = mk[0].data_ptr() + mk[1], mk[2]
        rc = lib().npp_affine_add_fin_multi_x(C.cast(jobs, C.c_void_p), n, xch, stream_ptr())
        if rc == L.NPP_E_UNSUPPORTED and xch < 0:
            return False
        check(rc, "npp_affine_add_fin_multi_x")
        MULTI_STATS[0] += 1
        if xch >= 0:
            FOLD_STATS[0] += 1
        subs, saved = [], []
        for k in range(n):
            a, _ga, _ba, b, _gb, _bb = tens[6 * k:6 * k + 6]
            sa, sb, relu, _training, _out, _mk = metas[k]
            pa, pb, y = preps[k]
            sub = _SubCtx()
            sub.gslots = (sa.gslot, sb.gslot if sb is not None else None)
            _fin_record(sub, a, b, y, pa, pb, sa, sb, relu)
            subs.append((sub.relu, sub.sides, sub.gslots, sub.ticket))
            saved += list(sub.saved_tensors)
            outs.append(y)
        ctx.save_for_backward(*saved)
        ctx.subs = subs
        return True

    @staticmethod
    def _backward_multi(ctx, douts):
        """The n backward passes as ONE reduce launch + ONE apply launch when every item is the same kind of fused local BatchNorm
        backward (both operands BatchNorms, or one BatchNorm + nothing / a plain operand) on one shape; None otherwise (nothing launched)."""
        n = ctx.n
        saved = ctx.saved_tensors
        needs = ctx.needs_input_grad[1:]
        items = []
        kind = None
        sync_grp = None
        for k, dout in enumerate(douts):
            if dout is None:
                return None
            a, b, yrelu, mia, mib, _ssa, _ssb = saved[7 * k:7 * k + 7]
            bna, bnb, batch_a, batch_b, cnt_a, cnt_b, has_b = ctx.subs[k][1]
            ni = needs[6 * k:6 * k + 6]
            if bna is None or not batch_a or not ni[0] or mia is None:
                return None
            grp_k = _sync_group(bna)[0]
            two = bool(has_b and bnb is not None)
            if two:
                if not (batch_b and ni[3] and a.shape == b.shape and a.dtype == b.dtype and cnt_a == cnt_b and mib is not None
                        and _sync_group(bnb)[0] is grp_k and _fused_layout_ok(b)):
                    return None
            if k == 0:
                sync_grp = grp_k
            elif grp_k is not sync_grp:
                return None
            elif has_b and yrelu is not None:
                return None      # (a plain operand behind a ReLU needs a masked copy of its own)
            dout = to_nhwc(dout)
            if dout.dtype != a.dtype:
                dout = cast(dout, a.dtype)
            if not (_fused_layout_ok(a) and _fused_layout_ok(dout) and (yrelu is None or _fused_layout_ok(yrelu))):
                return None
            sig = (two, yrelu is not None, tuple(a.shape), a.dtype)
            if kind is None:
                kind = sig
            elif kind != sig:
                return None
            items.append((dout, a, b, yrelu, mia, mib, bna, bnb, cnt_a, has_b, ni, two))
        if not FUSE_BN_FIN:
            return None
        two = kind[0]
        dev = items[0][1].device
        c = items[0][1].shape[1]
        if sync_grp is not None:
            # SyncBatchNorm (every item of ONE group): the multi-job reduce, ONE peer-to-peer slab exchange with a segment per job
            # (it writes the LOCAL dgamma / dbeta -- DDP averages them afterwards, as torch.nn.SyncBatchNorm does -- leaves the world's
            # sums in replica 0 and zeroes the other replicas), the multi-job apply with the world's pixel count
            from . import comm
            if not (P2P_DIRECT and FUSE_BN_SYNC and BN_MULTI_SYNC and dev.type == "cuda" and n <= 8
                    and comm.p2p_can(n * (3 if two else 2) * c, sync_grp)):
                return None
        xch = _fold_channel(n * (3 if two else 2) * c, sync_grp) if sync_grp is not None else -1      # the exchange inside the apply launch
        jobs = (L.NppBnBwdJob * n)()
        segs = []
        keep, results, need_reduce = [], [], []
        for k, (dout, a, b, yrelu, mia, mib, bna, bnb, cnt, has_b, ni, _two) in enumerate(items):
            gs = ctx.subs[k][2]
            tk_ = ctx.subs[k][3] if len(ctx.subs[k]) > 3 else None
            sums = tk_.take(dout, two) if tk_ is not None else None      # delivered by the gradient's last writer: no reduce for this job
            need_reduce.append(sums is None)
            if sums is None:
                sums = zeros_f64(R * (3 if two else 2) * c, dev)
            j = jobs[k]
            j.dout, j.ya = desc(dout), desc(a)
            j.yb = desc(b) if two else _zero_desc()
            j.relu_out = desc(yrelu) if yrelu is not None else _zero_desc()
            dxa = _dx_into(gs[0], a)
            j.dya = desc(dxa)
            dxb = None
            if two:
                dxb = _dx_into(gs[1], b)
                j.dyb = desc(dxb)
            else:
                j.dyb = _zero_desc()
            j.mi_a = mia.data_ptr()
            ga = bna.weight.detach() if bna.weight is not None else None
            j.gamma_a = ptr(ga)
            dga = _grad_buf(bna.weight if ni[1] else None, c, dev)
            dba = _grad_buf(bna.bias if ni[2] else None, c, dev)
            if sync_grp is None or xch >= 0:
                j.dgamma_a, j.dbeta_a = dga.data_ptr(), dba.data_ptr()
            dgb = dbb = gb = None
            if two:
                j.mi_b = mib.data_ptr()
                gb = bnb.weight.detach() if bnb.weight is not None else None
                j.gamma_b = ptr(gb)
                dgb = _grad_buf(bnb.weight if ni[4] else None, c, dev)
                dbb = _grad_buf(bnb.bias if ni[5] else None, c, dev)
                if sync_grp is None or xch >= 0:
                    j.dgamma_b, j.dbeta_b = dgb.data_ptr(), dbb.data_ptr()
            if sync_grp is not None and xch < 0:      # [0, c): sum dout -> dbeta (of both sides); [c, 2c) -> dgamma_a; [2c, 3c) -> dgamma_b
                segs.append((sums, (3 if two else 2) * c, R, c, (dba, dbb, dga, dgb), True))
            j.sums, j.count = sums.data_ptr(), float(cnt)
            keep.append((sums, ga, gb))
            if two:
                results.append((dxa, dga if ni[1] else None, dba if ni[2] else None, dxb, dgb if ni[4] else None, dbb if ni[5] else None))
            else:
                passed = None
                if has_b and ni[3]:      # the plain operand receives the incoming gradient itself: several readers from here on
                    try:
                        dout._npp_own = False
                    except Exception:      # noqa: BLE001
                        pass
                    passed = dout
                results.append((dxa, dga if ni[1] else None, dba if ni[2] else None, passed, None, None))
        nb = lib().npp_reduce_blocks(items[0][1].shape[0] * items[0][1].shape[2] * items[0][1].shape[3], c, L.npp_dtype(items[0][1].dtype))
        s = stream_ptr()
        nr = sum(need_reduce)
        if nr == n:
            rjobs = jobs
        else:
            rjobs = (L.NppBnBwdJob * max(nr, 1))()
            q = 0
            for k in range(n):
                if need_reduce[k]:
                    C.memmove(C.byref(rjobs[q]), C.byref(jobs[k]), C.sizeof(L.NppBnBwdJob))
                    q += 1
        if nr > 0:
            rc = lib().npp_bn_bwd_reduce_multi(C.cast(rjobs, C.c_void_p), nr, nb, s)
            if rc == L.NPP_E_UNSUPPORTED:
                if nr != n:
                    raise RuntimeError("npp_bn_bwd_reduce_multi refused a subset of jobs it takes as a whole")
                return None
            check(rc, "npp_bn_bwd_reduce_multi")
            MULTI_STATS[1] += 1
        if sync_grp is not None and xch < 0:
            if not comm.p2p_exchange_slabs(segs, sync_grp):
                raise RuntimeError("peer-to-peer exchange refused vectors it had accepted the size of")
            MULTI_STATS[3] += 1
        check(lib().npp_bn_bwd_apply_multi_x(C.cast(jobs, C.c_void_p), n, xch, s), "npp_bn_bwd_apply_multi_x")
        MULTI_STATS[2] += 1
        if xch >= 0:
            FOLD_STATS[1] += 1
        out = (None,)
        for r in results:
            out += r
        return out + (None,) * n

    @staticmethod
    def backward(ctx, *douts):
        n = ctx.n
        if BN_MULTI and 2 <= n <= BN_MULTI_MAX:
            res_m = _BnAddPair._backward_multi(ctx, douts)
            if res_m is not None:
                return res_m
        saved = ctx.saved_tensors
        needs = ctx.needs_input_grad[1:]
        gens, reqs, res = [], [None] * n, [None] * n
        for k, d in enumerate(douts):
            sub = _SubCtx(saved[7 * k:7 * k + 7], ctx.subs[k][0], ctx.subs[k][1], needs[6 * k:6 * k + 6], ctx.subs[k][2], ctx.subs[k][3])
            gens.append(_BnAdd._backward_gen(sub, d))
        for k in range(n):      # every reduce first ...
            try:
                reqs[k] = next(gens[k])
            except StopIteration as done:
                res[k] = done.value
        from . import comm
        # ... then the requested exchanges, merged: consecutive requests of one group, at most 8 segments per launch
        answers = [None] * n
        asking = [k for k in range(n) if reqs[k] is not None]
        i = 0
        while i < len(asking):
            grp = reqs[asking[i]][1]
            batch, nseg = [], 0
            while i < len(asking) and reqs[asking[i]][1] is grp and nseg + len(reqs[asking[i]][0]) <= 8:
                batch.append(asking[i])
                nseg += len(reqs[asking[i]][0])
                i += 1
            if not batch:      # (a single request of more than 8 segments: leave it to its own call below)
                batch = [asking[i]]
                i += 1
            took = False
            if len(batch) > 1:
                took = comm.p2p_exchange_slabs([sg for k in batch for sg in reqs[k][0]], grp)
            for k in batch:
                answers[k] = True if took else comm.p2p_exchange_slabs(reqs[k][0], reqs[k][1])
        for k in asking:         # ... then every apply
            try:
                gens[k].send(answers[k])
                raise RuntimeError("_BnAdd._backward_gen asked for a second exchange")
            except StopIteration as done:
                res[k] = done.value
        out = (None,)
        for k in range(n):
            out += tuple(res[k][:6])
        return out + (None,) * n


def _bn_add_prepare(sa, sb, relu, training, out):
    """The argument preparation of bn_add: (tensor args, meta) for _BnAdd / _BnAddPair."""
    a = to_nhwc(sa.x if sa.private else take(sa.x))
    sa.x = a
    ga = sa.bn.weight if sa.bn is not None else None
    ba = sa.bn.bias if sa.bn is not None else None
    b = gb = bb = None
    if sb is not None:
        b = to_nhwc(sb.x if sb.private else take(sb.x))
        sb.x = b
        if b.dtype != a.dtype:
            b = cast(b, a.dtype)
            sb.x = b
        if sb.bn is not None:
            gb, bb = sb.bn.weight, sb.bn.bias
    holder = mk = None
    if out is not None:
        y = out(a)
        if y is not None:
            holder = [y]
            mk = out.mask_spec() if hasattr(out, "mask_spec") else None
    if mk is None and holder is None and _mask_wanted(a):
        n, c, h, w = a.shape
        mk = (torch.empty(n * h * w * (c // 8), dtype=torch.uint8, device=a.device), 0, c // 8)
    return (a, ga, ba, b, gb, bb), (sa, sb, bool(relu), bool(training), holder, mk)


def bn_add_multi(specs):
    """[bn_add(*spec) for spec in specs] (spec = (sa, sb, relu, training, out)) as one autograd node, see _BnAddPair."""
    prep = [_bn_add_prepare(*sp) for sp in specs]
    tens = [t for ts, _m in prep for t in ts]
    res = _BnAddPair.apply(len(prep), *tens, *[m for _ts, m in prep])
    for r, (_ts, m) in zip(res, prep):
        _register_mask(r, m[5])
        if m[0].ticket is not None:
            r._npp_ticket, m[0].ticket = m[0].ticket, None
        if FAN_CENSUS is not None:
            r._npp_bn = True
    return list(res)


def bn_add_pair(spec0, spec1):
    r0, r1 = bn_add_multi([spec0, spec1])
    return r0, r1


def bn_add(sa: BnSide, sb: Optional[BnSide] = None, relu: bool = False, training: bool = True, out=None):
    """Fused BN-apply (+ second operand, BN'd or plain) (+ ReLU).  out: a ConcatBuffer slot (callable shape -> tensor or
    None) to write the result into instead of a fresh tensor."""
    a = to_nhwc(sa.x if sa.private else take(sa.x))
    sa.x = a
    ga = sa.bn.weight if sa.bn is not None else None
    ba = sa.bn.bias if sa.bn is not None else None
    b = gb = bb = None
    if sb is not None:
        b = to_nhwc(sb.x if sb.private else take(sb.x))
        sb.x = b
        if b.dtype != a.dtype:
            b = cast(b, a.dtype)
            sb.x = b
        if sb.bn is not None:
            gb, bb = sb.bn.weight, sb.bn.bias
    holder = mk = None
    if out is not None:
        y = out(a)
        if y is not None:
            holder = [y]
            mk = out.mask_spec() if hasattr(out, "mask_spec") else None
    if mk is None and holder is None and _mask_wanted(a):
        n, c, h, w = a.shape
        mk = (torch.empty(n * h * w * (c // 8), dtype=torch.uint8, device=a.device), 0, c // 8)
    res = _BnAdd.apply(a, ga, ba, b, gb, bb, sa, sb, bool(relu), bool(training), holder, mk)
    _register_mask(res, mk)
    if sa.ticket is not None:
        res._npp_ticket, sa.ticket = sa.ticket, None
    if FAN_CENSUS is not None:
        res._npp_bn = True
    return res


def add(a, b):
    return bn_add(BnSide(a), BnSide(b))


def channel_stats(x: torch.Tensor, level: int = 1) -> torch.Tensor:
    x = to_nhwc(x)
    st = stats_buffer(R * 2 * x.shape[1], x.device, level)
    check(lib().npp_channel_stats(_byref(x.detach()), st.data_ptr(), stream_ptr()), "npp_channel_stats")
    return st


# --------------------------------------------------------------------------------------------------
# pooling
# --------------------------------------------------------------------------------------------------
class _Pool3x3(Function):
    @staticmethod
    def forward(ctx, x, is_avg, stride, want_stats, fan=None):
        x = to_nhwc(x)
        ctx.fan = fan if x.dtype == torch.bfloat16 else None
        n, c, h, w = x.shape
        oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
        y = new_nhwc(n, c, oh, ow, x.dtype, x.device)
        amax = None if is_avg else torch.empty((n, oh, ow, c), dtype=torch.uint8, device=x.device)
        stats = stats_buffer(R * 2 * c, x.device, want_stats) if want_stats else None
        check(lib().npp_pool3x3_fwd(_byref(x), _byref(y), ptr(amax), int(is_avg), stride, ptr(stats), stream_ptr()),
              "npp_pool3x3_fwd")
        ctx.save_for_backward(amax)
        ctx.cfg = (is_avg, stride, tuple(x.shape), x.dtype)
        ctx.set_materialize_grads(False)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _):
        if dy is None:
            return None, None, None, None, None
        (amax,) = ctx.saved_tensors
        is_avg, stride, xshape, dtype = ctx.cfg
        dy = to_nhwc(dy)
        if dy.dtype != dtype:
            dy = cast(dy, dtype)
        if ctx.fan is not None:
            dx, accumulate = ctx.fan.claim(_Like(xshape, dtype, dy.device))
        else:
            dx, accumulate = None, False
        if dx is None:
            dx, accumulate = new_nhwc(*xshape, dtype, dy.device), False
        check(lib().npp_pool3x3_bwd_acc(_byref(dy), ptr(amax), _byref(dx), int(is_avg), stride, int(accumulate), stream_ptr()),
              "npp_pool3x3_bwd")
        return dx, None, None, None, None


def pool3x3(x, is_avg=False, stride=1, want_stats=False):
    xa, fan = take_acc(x)
    return _Pool3x3.apply(xa, bool(is_avg), int(stride), int(want_stats), fan)


class _Pool2x2(Function):
    @staticmethod
    def forward(ctx, x, is_avg, want_stats):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = new_nhwc(n, c, h // 2, w // 2, x.dtype, x.device)
        stats = stats_buffer(R * 2 * c, x.device, want_stats) if want_stats else None
        check(lib().npp_pool2x2_fwd(_byref(x), _byref(y), int(is_avg), ptr(stats), stream_ptr()), "npp_pool2x2_fwd")
        ctx.save_for_backward(None if is_avg else x)
        ctx.cfg = (is_avg, tuple(x.shape), x.dtype)
        ctx.set_materialize_grads(False)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _):
        if dy is None:
            return None, None, None
        (x,) = ctx.saved_tensors
        is_avg, xshape, dtype = ctx.cfg
        dy = to_nhwc(dy)
        if dy.dtype != dtype:
            dy = cast(dy, dtype)
        odd = xshape[2] % 2 or xshape[3] % 2
        if odd:
            raise NotImplementedError("pool2x2 backward with odd extents")
        dx = new_nhwc(*xshape, dtype, dy.device)
        check(lib().npp_pool2x2_bwd(_byref(dy), tref(x), _byref(dx), int(is_avg), stream_ptr()), "npp_pool2x2_bwd")
        return dx, None, None


def pool2x2(x, is_avg=True, want_stats=False):
    return _Pool2x2.apply(take(x), bool(is_avg), int(want_stats))


# --------------------------------------------------------------------------------------------------
# squeeze-excite gate:  y = x * sigmoid(W2 relu(W1 gap(x) + b1) + b2)
# --------------------------------------------------------------------------------------------------
SE_FUSED = os.environ.get("NPP_SE_FUSED", "1") != "0"      # 0: the seven-launch form of rounds 1-2 (float-atomic squeeze sums)
_pending_se_grads: list = []      # (pooled, hidden, dz, dw1, db1, dw2, db2 aliases, n, c, stream): parameter gradients of the SE gates
_se_grad_bufs = {"pin": None, "dev": None, "ev": None, "keep": []}


def _f32c(t):
    t = t.detach()
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


class _SEScale(Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, fan=None):
        x = to_nhwc(x)
        ctx.fan = fan if x.dtype == torch.bfloat16 else None
        n, c, h, w = x.shape
        dev = x.device
        s = stream_ptr()
        w1f, b1f, w2f, b2f = (_f32c(t) for t in (w1, b1, w2, b2))
        y = new_nhwc(n, c, h, w, x.dtype, dev)
        ctx.fused = SE_FUSED and bool(lib().npp_se_supported(c))
        if ctx.fused:
            # two launches: slab-wise squeeze sums, then gate MLP (in every workgroup's prologue) + scale (se.hip)
            st = torch.empty(n * (2 * c + c // 2), dtype=torch.float32, device=dev)
            pooled, hidden, gate = st[:n * c].view(n, c), st[n * c:n * c + n * (c // 2)].view(n, c // 2), st[n * c + n * (c // 2):].view(n, c)
            ws = torch.empty(int(lib().npp_se_ws_floats(n, c)), dtype=torch.float32, device=dev)
            check(lib().npp_se_fwd(_byref(x), w1f.data_ptr(), b1f.data_ptr(), w2f.data_ptr(), b2f.data_ptr(), _byref(y),
                                   pooled.data_ptr(), hidden.data_ptr(), gate.data_ptr(), ws.data_ptr(), s), "npp_se_fwd")
        else:
            pooled = zeros_f32(n * c, dev, own=True).view(n, c)
            check(lib().npp_global_avgpool(_byref(x), pooled.data_ptr(), s), "npp_global_avgpool")
            hidden = torch.empty((n, c // 2), dtype=torch.float32, device=dev)
            gate = torch.empty((n, c), dtype=torch.float32, device=dev)
            check(lib().npp_se_gate_fwd(pooled.data_ptr(), w1f.data_ptr(), b1f.data_ptr(), w2f.data_ptr(), b2f.data_ptr(),
                                        hidden.data_ptr(), gate.data_ptr(), n, c, s), "npp_se_gate_fwd")
            check(lib().npp_scale_channels(_byref(x), gate.data_ptr(), _byref(y), s), "npp_scale_channels")
        ctx.save_for_backward(x, w1, w2, pooled, hidden, gate, b1, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, pooled, hidden, gate, b1, b2 = ctx.saved_tensors
        dy = to_nhwc(dy)
        if dy.dtype != x.dtype:
            dy = cast(dy, x.dtype)
        n, c, h, w = x.shape
        dev = x.device
        s = stream_ptr()
        dw1 = _grad_buf(w1, w1.numel(), dev).view(w1.shape)
        db1 = _grad_buf(b1, c // 2, dev)
        dw2 = _grad_buf(w2, w2.numel(), dev).view(w2.shape)
        db2 = _grad_buf(b2, c, dev)
        w1f, w2f = _f32c(w1), _f32c(w2)
        dx, accumulate = _claim_dx(ctx.fan if ctx.fused else None, x)
        if ctx.fused:
            dz = torch.empty((n, c + c // 2), dtype=torch.float32, device=dev)
            ws = torch.empty(int(lib().npp_se_ws_floats(n, c)), dtype=torch.float32, device=dev)
            check(lib().npp_se_bwd_acc(_byref(dy), _byref(x), w1f.data_ptr(), w2f.data_ptr(), hidden.data_ptr(), gate.data_ptr(),
                                       _byref(dx), dz.data_ptr(), ws.data_ptr(), int(accumulate), s), "npp_se_bwd")
            # parameter gradients: nobody reads them before the optimizer -> one batched launch per step under TrainStep
            if (DEFER_WGRAD_MAX_PIX > 0 and all(t.dtype == torch.float32 for t in (w1, b1, w2, b2))
                    and _may_defer(w1) and _may_defer(w2) and _may_defer(b1) and _may_defer(b2)):
                def alias(t):
                    return torch.empty(0, dtype=t.dtype, device=t.device).set_(t.untyped_storage(), t.storage_offset(), t.shape, t.stride())
                _pending_se_grads.append((pooled, hidden, dz, alias(dw1), alias(db1), alias(dw2), alias(db2), n, c,
                                          torch.cuda.current_stream()))
            else:
                it = L.NppSeGradItem(pooled.data_ptr(), hidden.data_ptr(), dz.data_ptr(), dw1.data_ptr(), db1.data_ptr(),
                                     dw2.data_ptr(), db2.data_ptr(), n, c)
                check(lib().npp_se_param_grads(C.byref(it), s), "npp_se_param_grads")
        else:
            dgate = zeros_f32(n * c, dev).view(n, c)
            check(lib().npp_se_bwd_reduce(_byref(dy), _byref(x), dgate.data_ptr(), s), "npp_se_bwd_reduce")
            dpooled = torch.empty((n, c), dtype=torch.float32, device=dev)
            scratch = torch.empty((n, c + c // 2), dtype=torch.float32, device=dev)
            check(lib().npp_se_gate_bwd(pooled.data_ptr(), hidden.data_ptr(), gate.data_ptr(), dgate.data_ptr(),
                                        w1f.data_ptr(), w2f.data_ptr(), dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(),
                                        db2.data_ptr(), dpooled.data_ptr(), scratch.data_ptr(), n, c, s), "npp_se_gate_bwd")
            check(lib().npp_se_bwd_apply(_byref(dy), gate.data_ptr(), dpooled.data_ptr(), _byref(dx), s), "npp_se_bwd_apply")
        if dw1.dtype != w1.dtype:
            dw1, db1, dw2, db2 = dw1.to(w1.dtype), db1.to(b1.dtype), dw2.to(w2.dtype), db2.to(b2.dtype)
        return dx, dw1, db1, dw2, db2, None


def _flush_se_grads():
    if not _pending_se_grads:
        return
    items = list(_pending_se_grads)
    _pending_se_grads.clear()
    cur = torch.cuda.current_stream()
    seen = {cur.cuda_stream}
    for it in items:
        if it[9].cuda_stream not in seen:
            seen.add(it[9].cuda_stream)
            cur.wait_stream(it[9])
    n = len(items)
    arr = (L.NppSeGradItem * n)()
    for i, (pooled, hidden, dz, dw1, db1, dw2, db2, nimg, c, _st) in enumerate(items):
        arr[i] = L.NppSeGradItem(pooled.data_ptr(), hidden.data_ptr(), dz.data_ptr(), dw1.data_ptr(), db1.data_ptr(),
                                 dw2.data_ptr(), db2.data_ptr(), nimg, c)
    nb = int(lib().npp_se_param_grads_batched_ws(C.cast(arr, C.c_void_p), n))
    capturing = torch.cuda.is_current_stream_capturing()
    bufs = _table_bufs(_se_grad_bufs, nb, items[0][0].device, capturing) if nb > 0 else None
    if bufs is None:
        for i in range(n):
            check(lib().npp_se_param_grads(C.byref(arr[i]), stream_ptr()), "npp_se_param_grads")
    else:
        pin, dev = bufs
        check(lib().npp_se_param_grads_batched(C.cast(arr, C.c_void_p), n, pin.data_ptr(), dev.data_ptr(), pin.numel(), stream_ptr()),
              "npp_se_param_grads_batched")
        if not capturing:
            _se_grad_bufs["ev"] = torch.cuda.Event()
            _se_grad_bufs["ev"].record()
    for it in items:
        if it[9].cuda_stream != cur.cuda_stream:
            for t in it[:7]:
                t.record_stream(cur)


def se_scale(x, w1, b1, w2, b2):
    xa, fan = take_acc(x)
    return _SEScale.apply(xa, w1, b1, w2, b2, fan)


class _SEScale2(Function):
    """Two squeeze-excite gates with their own weights on ONE tensor (ENCODER.normal / .reduce apply `se_connect` to state 1 twice,
    genotypes.py:30-36): one squeeze pass + one gate-and-scale launch for both; backward one partial-sum launch and one apply launch
    that writes the SUM of the two gradients w.r.t. x (npp_se_fwd_multi / npp_se_bwd_multi).  Args: x, then (w1, b1, w2, b2) x 2."""

    @staticmethod
    def forward(ctx, x, fan, *ws):
        x = to_nhwc(x)
        ctx.fan = fan if x.dtype == torch.bfloat16 else None
        n, c, h, w = x.shape
        dev = x.device
        jobs = (L.NppSeFwdJob * 2)()
        ys, keep, saved = [], [], []
        for q in range(2):
            w1, b1, w2, b2 = ws[4 * q:4 * q + 4]
            w1f, b1f, w2f, b2f = (_f32c(t) for t in (w1, b1, w2, b2))
            y = new_nhwc(n, c, h, w, x.dtype, dev)
            st = torch.empty(n * (2 * c + c // 2), dtype=torch.float32, device=dev)
            pooled, hidden, gate = st[:n * c].view(n, c), st[n * c:n * c + n * (c // 2)].view(n, c // 2), st[n * c + n * (c // 2):].view(n, c)
            jobs[q] = L.NppSeFwdJob(desc(y), w1f.data_ptr(), b1f.data_ptr(), w2f.data_ptr(), b2f.data_ptr(), pooled.data_ptr(),
                                    hidden.data_ptr(), gate.data_ptr())
            ys.append(y)
            keep.append((w1f, b1f, w2f, b2f))
            saved += [w1, w2, pooled, hidden, gate, b1, b2]
        wsb = torch.empty(int(lib().npp_se_ws_floats(n, c)), dtype=torch.float32, device=dev)
        check(lib().npp_se_fwd_multi(_byref(x), C.cast(jobs, C.c_void_p), 2, wsb.data_ptr(), stream_ptr()), "npp_se_fwd_multi")
        ctx.save_for_backward(x, *saved)
        ctx.set_materialize_grads(False)
        return tuple(ys)

    @staticmethod
    def backward(ctx, dy0, dy1):
        x, *rest = ctx.saved_tensors
        n, c, h, w = x.shape
        dev = x.device
        s = stream_ptr()
        sets = [rest[7 * q:7 * q + 7] for q in range(2)]      # w1, w2, pooled, hidden, gate, b1, b2
        dys = []
        for d in (dy0, dy1):
            if d is None:
                d = new_nhwc(n, c, h, w, x.dtype, dev, zero=True)
            d = to_nhwc(d)
            dys.append(d if d.dtype == x.dtype else cast(d, x.dtype))
        dx, accumulate = _claim_dx(ctx.fan, x)
        jobs = (L.NppSeBwdJob * 2)()
        keep, grads, dzs = [], [], []
        for q in range(2):
            w1, w2, pooled, hidden, gate, b1, b2 = sets[q]
            w1f, w2f = _f32c(w1), _f32c(w2)
            dz = torch.empty((n, c + c // 2), dtype=torch.float32, device=dev)
            jobs[q] = L.NppSeBwdJob(desc(dys[q]), w1f.data_ptr(), w2f.data_ptr(), hidden.data_ptr(), gate.data_ptr(), dz.data_ptr())
            keep.append((w1f, w2f))
            dzs.append(dz)
        wsb = torch.empty(2 * int(lib().npp_se_ws_floats(n, c)), dtype=torch.float32, device=dev)
        check(lib().npp_se_bwd_multi(_byref(x), C.cast(jobs, C.c_void_p), 2, _byref(dx), wsb.data_ptr(), int(accumulate), s),
              "npp_se_bwd_multi")
        for q in range(2):
            w1, w2, pooled, hidden, gate, b1, b2 = sets[q]
            dw1 = _grad_buf(w1, w1.numel(), dev).view(w1.shape)
            db1 = _grad_buf(b1, c // 2, dev)
            dw2 = _grad_buf(w2, w2.numel(), dev).view(w2.shape)
            db2 = _grad_buf(b2, c, dev)
            dz = dzs[q]
            if (DEFER_WGRAD_MAX_PIX > 0 and all(t.dtype == torch.float32 for t in (w1, b1, w2, b2))
                    and _may_defer(w1) and _may_defer(w2) and _may_defer(b1) and _may_defer(b2)):
                def alias(t):
                    return torch.empty(0, dtype=t.dtype, device=t.device).set_(t.untyped_storage(), t.storage_offset(), t.shape, t.stride())
                _pending_se_grads.append((pooled, hidden, dz, alias(dw1), alias(db1), alias(dw2), alias(db2), n, c,
                                          torch.cuda.current_stream()))
            else:
                it = L.NppSeGradItem(pooled.data_ptr(), hidden.data_ptr(), dz.data_ptr(), dw1.data_ptr(), db1.data_ptr(),
                                     dw2.data_ptr(), db2.data_ptr(), n, c)
                check(lib().npp_se_param_grads(C.byref(it), s), "npp_se_param_grads")
            if dw1.dtype != w1.dtype:
                dw1, db1, dw2, db2 = dw1.to(w1.dtype), db1.to(b1.dtype), dw2.to(w2.dtype), db2.to(b2.dtype)
            grads += [dw1, db1, dw2, db2]
        return (dx, None, *grads)


SE_PAIR = os.environ.get("NPP_SE_PAIR", "1") != "0"
SE_PAIR_STATS = [0]


def se_scale_pair(x, params):
    """params: [(w1, b1, w2, b2)] x 2 -> (y0, y1)."""
    xa, fan = take_acc(x)
    SE_PAIR_STATS[0] += 1
    return _SEScale2.apply(xa, fan, *params[0], *params[1])


# --------------------------------------------------------------------------------------------------
# bilinear (align_corners=True)
# --------------------------------------------------------------------------------------------------
class _Bilinear(Function):
    @staticmethod
    def forward(ctx, x, oh, ow, align_corners):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = new_nhwc(n, c, oh, ow, x.dtype, x.device)
        check(lib().npp_bilinear_fwd_ac(_byref(x), _byref(y), int(align_corners), stream_ptr()), "npp_bilinear_fwd")
        ctx.cfg = (tuple(x.shape), x.dtype, int(align_corners))
        return y

    @staticmethod
    def backward(ctx, dy):
        xshape, dtype, ac = ctx.cfg
        dy = to_nhwc(dy)
        if dy.dtype != dtype:
            dy = cast(dy, dtype)
        dx = new_nhwc(*xshape, dtype, dy.device)
        _bilinear_bwd(dy, dx, ac)
        return dx, None, None, None


def _bilinear_bwd(dy, dx, ac):
    """npp_bilinear_bwd_ws: the transpose of the interpolation, as two 1-D passes through scratch when the library asks for it."""
    nb = int(lib().npp_bilinear_bwd_ws_bytes(_byref(dy), _byref(dx)))
    if nb:
        ws = torch.empty(nb, dtype=torch.uint8, device=dy.device)
        check(lib().npp_bilinear_bwd_ws(_byref(dy), _byref(dx), int(ac), ws.data_ptr(), nb, stream_ptr()), "npp_bilinear_bwd_ws")
    else:
        check(lib().npp_bilinear_bwd_ac(_byref(dy), _byref(dx), int(ac), stream_ptr()), "npp_bilinear_bwd")


def bilinear(x, oh: int, ow: int, align_corners: bool = True):
    if x.shape[2] == oh and x.shape[3] == ow:
        return x   # identity resample (Interpolate(1.0), model_augment.py:638-645) is exact with either convention
    return _Bilinear.apply(take(x), int(oh), int(ow), bool(align_corners))


def interpolate_scale(x, scale: float):
    """F.interpolate(x, scale_factor=scale, mode='bilinear', align_corners=True): out = floor(in * scale)."""
    import math
    oh, ow = int(math.floor(x.shape[2] * scale)), int(math.floor(x.shape[3] * scale))
    return bilinear(x, oh, ow)


# --------------------------------------------------------------------------------------------------
# layout plumbing
# --------------------------------------------------------------------------------------------------
class _Cast(Function):
    @staticmethod
    def forward(ctx, x, dtype):
        x = to_nhwc(x)
        y = new_nhwc(*x.shape, dtype, x.device)
        check(lib().npp_copy(_byref(x), _byref(y), stream_ptr()), "npp_copy")
        ctx.dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        return _Cast.apply(dy, ctx.dtype), None


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    return _Cast.apply(x, dtype)


class _Concat(Function):
    @staticmethod
    def forward(ctx, want_mask, *xs):
        xs = [to_nhwc(x) for x in xs]
        n, _, h, w = xs[0].shape
        ctot = sum(x.shape[1] for x in xs)
        y = new_nhwc(n, ctot, h, w, xs[0].dtype, xs[0].device)
        s = stream_ptr()
        if len(xs) <= 8 and all(x.dtype == y.dtype for x in xs):
            descs = [desc(x) for x in xs]
            arr = (C.POINTER(L.NppTensor) * len(xs))(*[C.pointer(d) for d in descs])
            if want_mask and all(x.shape[1] % 8 == 0 and L.nhwc_ld(x) % 8 == 0 for x in xs):
                mk = torch.empty(n * h * w * (ctot // 8), dtype=torch.uint8, device=y.device)
                check(lib().npp_concat_m(arr, len(xs), _byref(y), mk.data_ptr(), ctot // 8, s), "npp_concat_m")
                _Concat.last_mask = (mk, 0, ctot // 8)
            else:
                _Concat.last_mask = None
                check(lib().npp_concat(arr, len(xs), _byref(y), s), "npp_concat")
        else:
            off = 0
            for x in xs:
                c = x.shape[1]
                check(lib().npp_copy(_byref(x), _byref(y[:, off:off + c]), s), "npp_copy")
                off += c
        ctx.splits = [x.shape[1] for x in xs]
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = to_nhwc(dy)
        outs = [None]
        off = 0
        for c in ctx.splits:
            outs.append(dy[:, off:off + c])   # channel-slice views: consumers take ld, no copy
            off += c
        return tuple(outs)


def concat(xs: Sequence[torch.Tensor]):
    _Concat.last_mask = None
    res = _Concat.apply(_mask_wanted(xs[0]), *[take(x) for x in xs])
    _register_mask(res, _Concat.last_mask)
    _Concat.last_mask = None
    return res


class _SplitHalf(Function):
    """x -> (x[:, :C/2], x[:, C/2:]) as channel-slice views (the PC-DARTS partial-channel split, model_search_interact.py:59-60).
    autograd's own slicing materialises, per half, a full-size zero tensor + a copy and then adds the two (5 full passes and
    5 launches per MixedOp); here the backward is ONE npp_concat of the two half gradients."""

    @staticmethod
    def forward(ctx, x):
        x = to_nhwc(x)
        c = x.shape[1]
        ctx.c = c
        ctx.set_materialize_grads(False)
        return _alias_slice(x, 0, c // 2), _alias_slice(x, c // 2, c - c // 2)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None and g2 is None:
            return None
        ref = g1 if g1 is not None else g2
        n, _, h, w = ref.shape
        c = ctx.c
        parts = []
        for g, cc in ((g1, c // 2), (g2, c - c // 2)):
            parts.append(to_nhwc(g) if g is not None else new_nhwc(n, cc, h, w, ref.dtype, ref.device, zero=True))
        if parts[0].dtype != parts[1].dtype:
            parts[1] = cast(parts[1], parts[0].dtype)
        y = new_nhwc(n, c, h, w, parts[0].dtype, ref.device)
        descs = [desc(t) for t in parts]
        arr = (C.POINTER(L.NppTensor) * 2)(*[C.pointer(d) for d in descs])
        check(lib().npp_concat(arr, 2, _byref(y), stream_ptr()), "npp_concat")
        return y


def _alias_slice(x: torch.Tensor, c0: int, c: int) -> torch.Tensor:
    """Channels [c0, c0 + c) of the NHWC tensor x as a tensor of its own (same storage; see _alias)."""
    n, _, h, w = x.shape
    t = torch.empty(0, dtype=x.dtype, device=x.device)
    return t.set_(x.untyped_storage(), x.storage_offset() + c0 * x.stride(1), (n, c, h, w), x.stride())


def split_half(x):
    """(x[:, :C/2], x[:, C/2:]) without autograd's slice-backward passes."""
    return _SplitHalf.apply(take(x))


def _alias(base: torch.Tensor, c0: int, c: int) -> torch.Tensor:
    """Channels [c0, c0 + c) of the NHWC buffer `base` as a tensor of its own (same storage, no view relation: autograd
    must not see the producers' writes as in-place updates of a view of the buffer)."""
    n, _, h, w = base.shape
    sn, sc, sh, sw = base.stride()
    t = torch.empty(0, dtype=base.dtype, device=base.device)
    return t.set_(base.untyped_storage(), base.storage_offset() + c0, (n, c, h, w), (sn, sc, sh, sw))


class _ConcatAlias(Function):
    """The cell output whose parts already sit in their channel slices: no data moves forward, backward hands every part
    its slice of the gradient (views)."""

    @staticmethod
    def forward(ctx, holder, *parts):
        buf = holder[0]
        ctx.splits = [x.shape[1] for x in parts]
        ctx.accs = holder[1] if len(holder) > 1 else None      # the parts' shared gradient buffers (_FanAcc), see backward
        return _alias(buf, 0, buf.shape[1])

    @staticmethod
    def backward(ctx, dy):
        own = getattr(dy, "_npp_own", False) and FAN_ACCUM
        dy = to_nhwc(dy)
        outs, off = [None], 0
        for k, c in enumerate(ctx.splits):
            v = dy[:, off:off + c]
            # the concatenation is the LAST consumer of its parts, so this runs before their other consumers' backward: an owned
            # gradient's slice becomes the part's shared buffer and the convs that read the part add into it (no add_n for the slice)
            if own and ctx.accs is not None and ctx.accs[k] is not None:
                ctx.accs[k].seed(v)
            outs.append(v)
            off += c
        return tuple(outs)


class ConcatBuffer:
    """torch.cat(states, dim=1) of a cell (model_augment.py:62, 106, 172, 229) without the copy: the buffer of the
    concatenated output is allocated when the first part is about to be produced, every part is WRITTEN into its channel
    slice by the kernel that produces it (bn_add's out=), and result() ties the buffer to the parts for autograd.  58
    concat launches and 5.8 GB of HBM traffic per training step (batch 16, 384 x 384) were exactly this copy.
    NPP_CONCAT_INPLACE=0 restores npp_concat."""

    ENABLED = os.environ.get("NPP_CONCAT_INPLACE", "1") != "0"
    MINC = int(os.environ.get("NPP_CONCAT_MINC", "0"))          # bisection aids: part widths that take the in-place path
    MAXC = int(os.environ.get("NPP_CONCAT_MAXC", "1000000"))

    def __init__(self, nparts: int):
        self.nparts = nparts
        self.buf = None
        self.parts: List[Optional[torch.Tensor]] = [None] * nparts
        self.c = None
        self.mask = None

    def slot(self, k: int):
        """out= argument for the producer of part k: slot(like) -> the tensor to write (or None), slot.mask_spec() -> where the
        part's ReLU bit-mask goes (the buffer owns ONE mask for the whole concatenation: consumers of a part and consumers
        of the whole both find their bits in it)."""
        cb = self

        class _Slot:
            def __call__(self, like: torch.Tensor):
                if not ConcatBuffer.ENABLED or like.dim() != 4:
                    return None
                n, c, h, w = like.shape
                if cb.buf is None:
                    if c % 8 != 0 or not (ConcatBuffer.MINC <= c <= ConcatBuffer.MAXC):
                        return None
                    cb.c = c
                    cb.buf = new_nhwc(n, c * cb.nparts, h, w, like.dtype, like.device)
                    if _mask_wanted(like):
                        cb.mask = torch.empty(n * h * w * (c * cb.nparts // 8), dtype=torch.uint8, device=like.device)
                b = cb.buf
                if (c != cb.c or b.shape[0] != n or b.shape[2] != h or b.shape[3] != w or b.dtype != like.dtype
                        or b.device != like.device):
                    return None
                return _alias(b, k * c, c)

            def mask_spec(self):
                if cb.mask is None:
                    return None
                return (cb.mask, k * cb.c // 8, cb.c * cb.nparts // 8)
        return _Slot()

    def result(self, parts: Sequence[torch.Tensor]) -> torch.Tensor:
        """The concatenation of `parts` (the tensors the producers returned, in order)."""
        ok = self.buf is not None and len(parts) == self.nparts
        if ok:
            base = self.buf.data_ptr()
            esz = self.buf.element_size()
            for k, t in enumerate(parts):
                if t.data_ptr() != base + k * self.c * esz or t.shape[1] != self.c:
                    ok = False
                    break
        if not ok:
            return concat(parts)
        taken = [take_acc(t) for t in parts]
        res = _ConcatAlias.apply([self.buf, [a for _, a in taken]], *[t for t, _ in taken])
        if self.mask is not None:
            _register_mask(res, (self.mask, 0, self.c * self.nparts // 8))
        return res


class _ImageToNhwc(Function):
    @staticmethod
    def forward(ctx, x, dtype, cpad):
        xf = x.detach()
        if xf.dtype != torch.float32 or not xf.is_contiguous():
            xf = xf.float().contiguous()
        n, c, h, w = xf.shape
        y = new_nhwc(n, cpad, h, w, dtype, x.device)
        check(lib().npp_nchw_to_nhwc(xf.data_ptr(), n, c, h, w, _byref(y), stream_ptr()), "npp_nchw_to_nhwc")
        return y[:, :c]

    @staticmethod
    def backward(ctx, dy):
        return None, None, None   # the network input needs no gradient (core/function.py:87)


def image_to_nhwc(x, dtype, cpad=8):
    """NCHW f32 image batch -> NHWC `dtype`, channels zero-padded to `cpad` in memory (logical C kept)."""
    return _ImageToNhwc.apply(x, dtype, cpad)


def nhwc_to_nchw_f32(x):
    x = to_nhwc(x)
    out = torch.empty(tuple(x.shape), dtype=torch.float32, device=x.device)
    check(lib().npp_nhwc_to_nchw(_byref(x.detach()), out.data_ptr(), stream_ptr()), "npp_nhwc_to_nchw")
    return out


# --------------------------------------------------------------------------------------------------
# loss heads
# --------------------------------------------------------------------------------------------------
class _MseSse(Function):
    """sum((w * (pred - target))^2) as an f32 scalar; target is f32 NCHW-contiguous, w (optional) f32 [N, C]."""

    @staticmethod
    def forward(ctx, pred, target, weight):
        pred = to_nhwc(pred)
        tgt = target.detach()
        if tgt.dtype != torch.float32 or not tgt.is_contiguous():
            tgt = tgt.float().contiguous()
        wt = None
        if weight is not None:
            wt = weight.detach().reshape(pred.shape[0], pred.shape[1]).float().contiguous()
        sse = torch.zeros(1, dtype=torch.float64, device=pred.device)
        check(lib().npp_mse_w_fwd(_byref(pred), tgt.data_ptr(), ptr(wt), sse.data_ptr(), stream_ptr()), "npp_mse_fwd")
        ctx.save_for_backward(pred, tgt, wt)
        return sse.float().reshape(())

    @staticmethod
    def backward(ctx, g):
        pred, tgt, wt = ctx.saved_tensors
        gs = g.detach().float().reshape(1).contiguous()
        grad = new_nhwc(*pred.shape, pred.dtype, pred.device)
        check(lib().npp_mse_w_bwd(_byref(pred), tgt.data_ptr(), ptr(wt), gs.data_ptr(), _byref(grad), stream_ptr()), "npp_mse_bwd")
        return grad, None, None


def mse_sse(pred, target, weight=None):
    return _MseSse.apply(take(pred), target, weight)


class _UpsampledCE(Function):
    """Cross-entropy of logits bilinearly upsampled (align_corners) to the label size.
    ohem=(thresh, min_kept): OhemCrossEntropy (core/criterion.py:54-72), mean over kept pixels;
    ohem=None: F.cross_entropy(weight, ignore_index) weighted mean (core/criterion.py:194-197)."""

    @staticmethod
    def forward(ctx, logits, labels, class_w, ignore, ohem):
        logits = to_nhwc(logits)
        n, c, h, w = logits.shape
        H, W = labels.shape[1], labels.shape[2]
        dev = logits.device
        lab = labels.detach()
        if lab.dtype != torch.int64 or not lab.is_contiguous():
            lab = lab.long().contiguous()
        cw = class_w.detach().float().contiguous()
        npx = n * H * W
        p_gt = torch.empty(npx, dtype=torch.float32, device=dev)
        wnll = torch.empty(npx, dtype=torch.float32, device=dev)
        s = stream_ptr()
        check(lib().npp_ce_pixel_fwd(_byref(logits), lab.data_ptr(), H, W, cw.data_ptr(), ignore, p_gt.data_ptr(),
                                     wnll.data_ptr(), s), "npp_ce_pixel_fwd")
        kth = None
        thresh = 0.0
        if ohem is not None:
            thresh, min_kept = ohem
            ws = torch.empty(260, dtype=torch.int32, device=dev)
            kth = torch.empty(2, dtype=torch.float32, device=dev)
            check(lib().npp_kth_smallest(p_gt.data_ptr(), npx, max(1, int(min_kept)), ws.data_ptr(), kth.data_ptr(), s),
                  "npp_kth_smallest")
        acc = torch.zeros(3, dtype=torch.float64, device=dev)
        check(lib().npp_ce_reduce(p_gt.data_ptr(), wnll.data_ptr(), lab.data_ptr(), cw.data_ptr(), ignore, npx, ptr(kth),
                                  float(thresh), int(ohem is not None), acc.data_ptr(), s), "npp_ce_reduce")
        denom = acc[1] if ohem is not None else acc[2]
        loss = (acc[0] / denom).float()
        ctx.save_for_backward(logits, lab, cw, p_gt, kth, denom)
        ctx.cfg = (ignore, thresh, ohem is not None, H, W)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, lab, cw, p_gt, kth, denom = ctx.saved_tensors
        ignore, thresh, use_ohem, H, W = ctx.cfg
        gs = (g.detach().double() / denom).float().reshape(1).contiguous()
        n, c, h, w = logits.shape
        s = stream_ptr()
        # gradient at the label resolution (f32 rows of c), then the atomic-free bilinear transpose down to h x w
        # (in the logits' own dtype: the bf16 image is 113 MB instead of 189 MB at 16 x 384 x 384 x 20 and the transpose hands
        # back bf16 directly -- this section runs alone, both branches wait for it; rows padded to 8 channels, padding zeroed)
        vec = 8 if logits.dtype == torch.bfloat16 else 4
        cpad = (c + vec - 1) // vec * vec          # whole 16-byte channel groups: the transpose then runs its vector kernel
        dup = new_nhwc(n, cpad, H, W, logits.dtype, logits.device)
        check(lib().npp_ce_pixel_grad_up_t(_byref(logits), lab.data_ptr(), cw.data_ptr(), ignore, p_gt.data_ptr(), ptr(kth),
                                           float(thresh), int(use_ohem), gs.data_ptr(), _byref(dup[:, :c]), s),
              "npp_ce_pixel_grad_up_t")
        dl = new_nhwc(n, cpad, h, w, logits.dtype, logits.device)
        _bilinear_bwd(dup, dl, 1)
        return (dl[:, :c] if cpad != c else dl), None, None, None, None


def upsampled_ce(logits, labels, class_w, ignore=255, ohem=None):
    return _UpsampledCE.apply(take(logits), labels, class_w, int(ignore), ohem)


# --------------------------------------------------------------------------------------------------
# a whole criterion as ONE autograd node: every per-term kernel + the scalar tail (npp_loss_tail_*), no 0-d ATen arithmetic
# --------------------------------------------------------------------------------------------------
FUSED_CRITERIA = os.environ.get("NPP_FUSED_CRITERIA", "1") != "0"


def _ce_term_fwd(logits, labels, class_w, ignore, ohem):
    """Forward kernels of one upsampled cross-entropy term; returns (acc f64[3] = [sum, kept, weight sum], den index, saved)."""
    logits = to_nhwc(logits)
    n, c, h, w = logits.shape
    H, W = labels.shape[1], labels.shape[2]
    dev = logits.device
    lab = labels.detach()
    if lab.dtype != torch.int64 or not lab.is_contiguous():
        lab = lab.long().contiguous()
    cw = class_w.detach()
    if cw.dtype != torch.float32 or not cw.is_contiguous():
        cw = cw.float().contiguous()
    npx = n * H * W
    p_gt = torch.empty(npx, dtype=torch.float32, device=dev)
    wnll = torch.empty(npx, dtype=torch.float32, device=dev)
    s = stream_ptr()
    check(lib().npp_ce_pixel_fwd(_byref(logits), lab.data_ptr(), H, W, cw.data_ptr(), ignore, p_gt.data_ptr(), wnll.data_ptr(), s),
          "npp_ce_pixel_fwd")
    kth = None
    thresh = 0.0
    if ohem is not None:
        thresh, min_kept = ohem
        ws = torch.empty(260, dtype=torch.int32, device=dev)
        kth = torch.empty(2, dtype=torch.float32, device=dev)
        check(lib().npp_kth_smallest(p_gt.data_ptr(), npx, max(1, int(min_kept)), ws.data_ptr(), kth.data_ptr(), s), "npp_kth_smallest")
    acc = zeros_f64(3, dev)
    check(lib().npp_ce_reduce(p_gt.data_ptr(), wnll.data_ptr(), lab.data_ptr(), cw.data_ptr(), ignore, npx, ptr(kth),
                              float(thresh), int(ohem is not None), acc.data_ptr(), s), "npp_ce_reduce")
    return acc, (1 if ohem is not None else 2), (logits, lab, cw, p_gt, kth, (ignore, thresh, ohem is not None, H, W))


def _ce_term_bwd(saved, gs_ptr):
    """Backward kernels of one cross-entropy term; gs_ptr: device f32 scalar = upstream gradient / denominator."""
    logits, lab, cw, p_gt, kth, (ignore, thresh, use_ohem, H, W) = saved
    n, c, h, w = logits.shape
    s = stream_ptr()
    vec = 8 if logits.dtype == torch.bfloat16 else 4
    cpad = (c + vec - 1) // vec * vec
    dup = new_nhwc(n, cpad, H, W, logits.dtype, logits.device)
    check(lib().npp_ce_pixel_grad_up_t(_byref(logits), lab.data_ptr(), cw.data_ptr(), ignore, p_gt.data_ptr(), ptr(kth),
                                       float(thresh), int(use_ohem), gs_ptr, _byref(dup[:, :c]), s), "npp_ce_pixel_grad_up_t")
    dl = new_nhwc(n, cpad, h, w, logits.dtype, logits.device)
    _bilinear_bwd(dup, dl, 1)
    return dl[:, :c] if cpad != c else dl


def _mse_term_fwd(pred, target, weight):
    pred = to_nhwc(pred)
    tgt = target.detach()
    if tgt.dtype != torch.float32 or not tgt.is_contiguous():
        tgt = tgt.float().contiguous()
    wt = None
    if weight is not None:
        wt = weight.detach().reshape(pred.shape[0], pred.shape[1]).float().contiguous()
    sse = zeros_f64(1, pred.device)
    check(lib().npp_mse_w_fwd(_byref(pred), tgt.data_ptr(), ptr(wt), sse.data_ptr(), stream_ptr()), "npp_mse_fwd")
    return sse, -1, (pred, tgt, wt)


def _mse_term_bwd(saved, gs_ptr):
    pred, tgt, wt = saved
    grad = new_nhwc(*pred.shape, pred.dtype, pred.device)
    check(lib().npp_mse_w_bwd(_byref(pred), tgt.data_ptr(), ptr(wt), gs_ptr, _byref(grad), stream_ptr()), "npp_mse_bwd")
    return grad


CRIT_SPLIT = os.environ.get("NPP_CRIT_SPLIT", "1") != "0"


class _nullctx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def _criterion_side(dev):
    """The second stream for the criteria's independent stage terms: the parsing branch's stream, while Network.forward runs its
    branches on two streams (hub stream set) -- never in the single-stream modes (NPP_STREAMS=1, the profiling leg)."""
    if not CRIT_SPLIT or dev.type != "cuda" or _hub_stream is None:
        return None
    for c in _stream_caches:
        if isinstance(c, dict):
            st = c.get((dev.type, dev.index, 0))
            if isinstance(st, torch.cuda.Stream) and st.cuda_stream != torch.cuda.current_stream().cuda_stream:
                return st
    return None


class _CriterionFused(Function):
    """loss = sum_i [ (sum_{t in stage i} coef_t * term_t) * exp(-lamda_i) + lamda_i ]  (core/criterion.py:139-142, 212-214).
    specs[t] = ("ce", labels, class_w, ignore, ohem, coef, stage) | ("mse", target, weight, coef, stage); xs[t] = the logits /
    heat-map tensor of term t."""

    @staticmethod
    def forward(ctx, lamda, specs, *xs):
        dev = xs[0].device
        nt, ns = len(xs), int(lamda.numel())
        terms = (L.NppLossTerm * nt)()
        saved = []
        keep = []
        # The terms of the refinement stages are independent (core/criterion.py:139-142, 212-214 sum them): the odd stages run on the
        # parsing branch's stream while the even ones run here -- the criteria sit between the end of forward and the start of backward
        # with nothing beside them (1.4 ms of the replayed step, one kernel in flight; round 4)
        side = _criterion_side(dev) if len({int(sp[6] if sp[0] == "ce" else sp[4]) % 2 for sp in specs}) > 1 else None
        cur = torch.cuda.current_stream() if side is not None else None
        if side is not None:
            side.wait_stream(cur)
        ctx.on_side = []
        for k, (sp, x) in enumerate(zip(specs, xs)):
            stage = int(sp[6] if sp[0] == "ce" else sp[4])
            there = side is not None and stage % 2 == 1
            ctx.on_side.append(there)
            if there:
                x.record_stream(side)
            with (torch.cuda.stream(side) if there else _nullctx()):
                if sp[0] == "ce":
                    acc, den, sv = _ce_term_fwd(x, sp[1], sp[2], sp[3], sp[4])
                    coef = sp[5]
                else:
                    acc, den, sv = _mse_term_fwd(x, sp[1], sp[2])
                    coef = sp[3]
            terms[k] = L.NppLossTerm(acc.data_ptr(), 0, den, float(coef), int(stage))
            saved.append((sp[0], sv))
            keep.append(acc)
        if side is not None:
            cur.wait_stream(side)
        lam = lamda.detach()
        if lam.dtype != torch.float32 or not lam.is_contiguous():
            lam = lam.float().contiguous()
        out = torch.empty(1 + nt + ns, dtype=torch.float32, device=dev)      # [loss | scales | unit]
        check(lib().npp_loss_tail_fwd(C.cast(terms, C.c_void_p), nt, lam.data_ptr(), ns, out.data_ptr(), out[1:].data_ptr(),
                                      out[1 + nt:].data_ptr(), stream_ptr()), "npp_loss_tail_fwd")
        ctx.terms = saved
        ctx.keep = keep
        ctx.tail = out
        ctx.counts = (nt, ns)
        ctx.lam_meta = (lamda.shape, lamda.dtype)
        return out[0]      # 0-d view of the buffer: no kernel

    @staticmethod
    def backward(ctx, g):
        nt, ns = ctx.counts
        out = ctx.tail
        gg = g.detach()
        if gg.dtype != torch.float32:
            gg = gg.float()
        gg = gg.reshape(1)
        if not gg.is_contiguous():
            gg = gg.contiguous()
        gbuf = torch.empty(nt + ns, dtype=torch.float32, device=out.device)      # [gs per term | dlamda]
        check(lib().npp_loss_tail_bwd(gg.data_ptr(), out[1:].data_ptr(), out[1 + nt:].data_ptr(), nt, ns, gbuf.data_ptr(),
                                      gbuf[nt:].data_ptr(), stream_ptr()), "npp_loss_tail_bwd")
        grads = []
        side = _criterion_side(out.device) if any(ctx.on_side) else None
        cur = torch.cuda.current_stream() if side is not None else None
        if side is not None:
            side.wait_stream(cur)          # (after the tail's backward: the terms read their scales from gbuf)
        for k, (kind, sv) in enumerate(ctx.terms):
            if not ctx.needs_input_grad[2 + k]:
                grads.append(None)
                continue
            gs_ptr = gbuf.data_ptr() + 4 * k
            there = side is not None and ctx.on_side[k]
            with (torch.cuda.stream(side) if there else _nullctx()):
                gk = _ce_term_bwd(sv, gs_ptr) if kind == "ce" else _mse_term_bwd(sv, gs_ptr)
            if there:
                gk.record_stream(cur)      # allocated on the side stream, consumed by the heads' backward elsewhere
            grads.append(gk)
        if side is not None:
            gbuf.record_stream(side)
            cur.wait_stream(side)
        dlam = gbuf[nt:].view(ctx.lam_meta[0])
        if dlam.dtype != ctx.lam_meta[1]:
            dlam = dlam.to(ctx.lam_meta[1])
        ctx.keep_g = gbuf
        return (dlam if ctx.needs_input_grad[0] else None, None) + tuple(grads)


def criterion_fused(lamda, specs, xs):
    return _CriterionFused.apply(lamda, tuple(specs), *[take(x) for x in xs])


def edge_class_weights_dev(labels: torch.Tensor) -> torch.Tensor:
    """edge_class_weights in two launches of the library (count + finish), no ATen arithmetic."""
    lab = labels.detach()
    if lab.dtype != torch.int64 or not lab.is_contiguous():
        lab = lab.long().contiguous()
    cnt = zeros_f64(2, lab.device)
    out = torch.empty(2, dtype=torch.float32, device=lab.device)
    check(lib().npp_edge_class_weights(lab.data_ptr(), lab.numel(), cnt.data_ptr(), out.data_ptr(), stream_ptr()), "npp_edge_class_weights")
    return out


def edge_class_weights(labels: torch.Tensor) -> torch.Tensor:
    """[pos/(pos+neg), neg/(pos+neg)] from the edge label map (core/criterion.py:161-166), on device."""
    lab = labels.detach()
    if lab.dtype != torch.int64 or not lab.is_contiguous():
        lab = lab.long().contiguous()
    cnt = torch.zeros(2, dtype=torch.float64, device=lab.device)
    check(lib().npp_edge_weights(lab.data_ptr(), lab.numel(), cnt.data_ptr(), stream_ptr()), "npp_edge_weights")
    tot = cnt.sum()
    return torch.stack([cnt[1] / tot, cnt[0] / tot]).float()


# --------------------------------------------------------------------------------------------------
# search supernet plumbing (model_search_interact.py:22-74)
# --------------------------------------------------------------------------------------------------
class _Nearest(Function):
    """F.interpolate(x, scale_factor=s) with the default mode 'nearest' (src = floor(dst / s))."""

    @staticmethod
    def forward(ctx, x, scale):
        import math
        x = to_nhwc(x)
        n, c, h, w = x.shape
        oh, ow = int(math.floor(h * scale)), int(math.floor(w * scale))
        y = new_nhwc(n, c, oh, ow, x.dtype, x.device)
        inv = 1.0 / float(scale)
        check(lib().npp_nearest(_byref(x), _byref(y), inv, inv, 0, stream_ptr()), "npp_nearest")
        ctx.cfg = (tuple(x.shape), x.dtype, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        xshape, dtype, inv = ctx.cfg
        dy = to_nhwc(dy)
        if dy.dtype != dtype:
            dy = cast(dy, dtype)
        dx = new_nhwc(*xshape, dtype, dy.device)
        check(lib().npp_nearest(_byref(dy), _byref(dx), inv, inv, 1, stream_ptr()), "npp_nearest(bwd)")
        return dx, None


def nearest(x, scale):
    if float(scale) == 1.0:
        return x
    return _Nearest.apply(take(x), float(scale))


class _WeightedSum(Function):
    """out = sum_k w[k] * y_k with w a 1-D tensor (softmaxed architecture weights); grads to every y_k and to w."""

    @staticmethod
    def forward(ctx, w, *ys):
        ys = [to_nhwc(y) for y in ys]
        k = len(ys)
        wf = w.detach().float().contiguous()
        out = new_nhwc(*ys[0].shape, ys[0].dtype, ys[0].device)
        descs = [desc(y) for y in ys]
        arr = (C.POINTER(L.NppTensor) * k)(*[C.pointer(d) for d in descs])
        check(lib().npp_weighted_sum_fwd(arr, k, wf.data_ptr(), _byref(out), stream_ptr()), "npp_weighted_sum_fwd")
        ctx.save_for_backward(wf, *ys)
        ctx.wdtype = w.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        wf, *ys = ctx.saved_tensors
        k = len(ys)
        dout = to_nhwc(dout)
        if dout.dtype != ys[0].dtype:
            dout = cast(dout, ys[0].dtype)
        need = ctx.needs_input_grad
        dys = [new_nhwc(*y.shape, y.dtype, y.device) if need[1 + i] else None for i, y in enumerate(ys)]
        ydesc = [desc(y) for y in ys]
        ddesc = [desc(d) if d is not None else None for d in dys]
        ya = (C.POINTER(L.NppTensor) * k)(*[C.pointer(d) for d in ydesc])
        da = (C.POINTER(L.NppTensor) * k)(*[C.pointer(d) if d is not None else None for d in ddesc])
        dw = zeros_f64(R * 8, dout.device)
        check(lib().npp_weighted_sum_bwd(ya, da, k, wf.data_ptr(), _byref(dout), dw.data_ptr(), stream_ptr()),
              "npp_weighted_sum_bwd")
        gw = None
        if need[0]:
            gw = torch.empty(8, dtype=torch.float32, device=dout.device)
            check(lib().npp_sum_replicas(dw.data_ptr(), R, 8, gw.data_ptr(), stream_ptr()), "npp_sum_replicas")
            gw = gw[:k] if ctx.wdtype == torch.float32 else gw[:k].to(ctx.wdtype)
        return (gw, *dys)


def weighted_sum(w, ys):
    return _WeightedSum.apply(w, *[take(y) for y in ys])


# The cells of the supernet index the softmax of the architecture parameters edge by edge (`weights[offset + j]`, `weights2[a:b]`:
# model_search_interact.py:1010-1020 in the reference).  Autograd gives every such index a node of its own whose backward is a zero
# fill of the whole matrix + a copy of the row + an add into the running sum: three launches per edge, ~650 per step of the supernet.
# One node per cell instead: the pieces are views handed out together, their gradients come back together and are laid side by side by
# ONE concatenation (pieces nobody used contribute zeros).
class _SplitPieces(Function):
    @staticmethod
    def forward(ctx, w, sizes):
        ctx.set_materialize_grads(False)
        ctx.sizes, ctx.shape, ctx.meta = sizes, w.shape, (w.dtype, w.device)
        outs, a = [], 0
        for n in sizes:
            outs.append(w[a] if n == 0 else w[a:a + n])      # n == 0: ONE row, as `w[a]` (a vector); n > 0: the slice w[a:a+n]
            a += max(n, 1)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        if all(g is None for g in gs):
            return None, None
        dtype, dev = ctx.meta
        tail = tuple(ctx.shape[1:])
        parts = []
        for n, g in zip(ctx.sizes, gs):
            rows = max(n, 1)
            if g is None:
                parts.append(torch.zeros((rows,) + tail, dtype=dtype, device=dev))
            else:
                parts.append(g.reshape((rows,) + tail).to(dtype))
        used = sum(max(n, 1) for n in ctx.sizes)
        if used < ctx.shape[0]:
            parts.append(torch.zeros((ctx.shape[0] - used,) + tail, dtype=dtype, device=dev))
        return torch.cat(parts, dim=0), None


def split_rows(w):
    """[w[0], w[1], ...] with one autograd node for all rows."""
    if not (isinstance(w, torch.Tensor) and w.requires_grad and torch.is_grad_enabled()):
        return [w[i] for i in range(w.shape[0])]
    return list(_SplitPieces.apply(w, (0,) * w.shape[0]))


def split_slices(w, sizes):
    """[w[0:s0], w[s0:s0+s1], ...] with one autograd node for all slices."""
    sizes = tuple(int(n) for n in sizes)
    if not (isinstance(w, torch.Tensor) and w.requires_grad and torch.is_grad_enabled()):
        outs, a = [], 0
        for n in sizes:
            outs.append(w[a:a + n]); a += n
        return outs
    return list(_SplitPieces.apply(w, sizes))


# ---- the mixed edge as one N-sided weighted BatchNorm sum (npp_mix_bn_fwd / npp_mix_bn_bwd) -----------------------------------------
MIX_FUSE = os.environ.get("NPP_MIX_FUSE", "1") != "0"


def _mix_side_struct(x, mi, bn, stats, track, dx=None):
    sd = L.NppMixSide()
    sd.x = desc(x)
    if dx is not None:
        sd.dx = desc(dx)
    if mi is not None:
        sd.mean_invstd = mi.data_ptr()
        sd.stats = stats.data_ptr() if stats is not None else None
        if track:
            sd.running_mean, sd.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            sd.num_batches_tracked = bn.num_batches_tracked.data_ptr() if bn.num_batches_tracked is not None else None
        sd.momentum = float(bn.momentum if bn.momentum is not None else 0.1)
        sd.eps = float(bn.eps)
    return sd


MIX_SYNC = os.environ.get("NPP_MIX_SYNC", "1") != "0"      # the fused mixed edge under SyncBatchNorm (peer-to-peer transport only)


def _mix_sync_group(bn, training: bool):
    """The process group of a SyncBatchNorm candidate that normalises with (exchanged) batch statistics, else None."""
    if bn is None or not (training or bn.running_mean is None):
        return None
    return _sync_group(bn)[0]


class _MixBnSum(Function):
    """out = sum_k w[k] * f_k(x_k), f_k = BatchNorm(affine=False, local batch statistics) of sides[k].bn or the identity."""

    @staticmethod
    def forward(ctx, w, sides, *xs):
        k = len(xs)
        dev = xs[0].device
        wf = w.detach().float().contiguous()
        out = new_nhwc(*xs[0].shape, xs[0].dtype, dev)
        c = xs[0].shape[1]
        arr = (L.NppMixSide * k)()
        mis, keep = [], []
        # SyncBatchNorm candidates: their statistics (all pending in this stream's pool) travel in ONE exchange, the kernel then
        # finalizes with the world's sample count
        sync_grp, sync_ws = None, 1
        for sd in sides:
            if sd.bn is not None and _sync_group(sd.bn)[0] is not None:
                sync_grp, sync_ws = _sync_group(sd.bn)
                break
        if sync_grp is not None:
            _presync_stats(sides, True)
        for i, (sd, x) in enumerate(zip(sides, xs)):
            mi = None
            if sd.bn is not None:
                if sd.stats is None:
                    sd.stats = channel_stats(x)
                mi = torch.empty(2 * c, dtype=torch.float32, device=dev)
                track = sd.bn.track_running_stats and sd.bn.running_mean is not None
                arr[i] = _mix_side_struct(x, mi, sd.bn, sd.stats, track)
                keep.append(sd.stats)
            else:
                arr[i] = _mix_side_struct(x, None, None, None, False)
            mis.append(mi)
        count = float(xs[0].shape[0] * xs[0].shape[2] * xs[0].shape[3]) * sync_ws
        if sync_grp is not None:
            check(lib().npp_mix_bn_fwd_n(C.cast(arr, C.c_void_p), k, wf.data_ptr(), _byref(out), count, stream_ptr()), "npp_mix_bn_fwd_n")
        else:
            check(lib().npp_mix_bn_fwd(C.cast(arr, C.c_void_p), k, wf.data_ptr(), _byref(out), stream_ptr()), "npp_mix_bn_fwd")
        ctx.sync = (sync_grp, count) if sync_grp is not None else None
        ctx.k = k
        ctx.bn_side = [m is not None for m in mis]
        ctx.wdtype = w.dtype
        ctx.save_for_backward(wf, *xs, *[m for m in mis if m is not None])
        return out

    @staticmethod
    def backward(ctx, dout):
        k = ctx.k
        saved = ctx.saved_tensors
        wf, xs, rest = saved[0], saved[1:1 + k], list(saved[1 + k:])
        mis = [rest.pop(0) if b else None for b in ctx.bn_side]
        dout = to_nhwc(dout)
        if dout.dtype != xs[0].dtype:
            dout = cast(dout, xs[0].dtype)
        dout = _gemm_ready(dout)
        need = ctx.needs_input_grad
        c = xs[0].shape[1]
        dxs = [new_nhwc(*x.shape, x.dtype, x.device) if need[2 + i] else None for i, x in enumerate(xs)]
        arr = (L.NppMixSide * k)()
        for i, x in enumerate(xs):
            sd = L.NppMixSide()
            sd.x = desc(x)
            if mis[i] is not None:
                sd.mean_invstd = mis[i].data_ptr()
            if dxs[i] is not None:
                sd.dx = desc(dxs[i])
            arr[i] = sd
        sums = zeros_f64(R * (k + 1) * c, dout.device)
        gw = torch.empty(8, dtype=torch.float32, device=dout.device)
        if ctx.sync is not None:
            # reduce -> the sums of the ranks through the mailboxes (replica 0 = the world's sums, the others zeroed; this rank's own
            # sums kept as floats for dw, which is NOT reduced) -> apply with the world's count
            from . import comm
            grp, count = ctx.sync
            check(lib().npp_mix_bn_bwd_reduce(C.cast(arr, C.c_void_p), k, _byref(dout), sums.data_ptr(), stream_ptr()), "npp_mix_bn_bwd_reduce")
            local = torch.empty((k + 1) * c, dtype=torch.float32, device=dout.device)
            if not comm.p2p_exchange_slabs([(sums, (k + 1) * c, R, c, (None, None, None, None, local), True)], grp):
                raise RuntimeError("_MixBnSum: the peer-to-peer exchange refused the sums it had accepted the size of in forward")
            check(lib().npp_mix_bn_bwd_apply(C.cast(arr, C.c_void_p), k, wf.data_ptr(), _byref(dout), sums.data_ptr(), count, local.data_ptr(),
                                             gw.data_ptr(), stream_ptr()), "npp_mix_bn_bwd_apply")
        else:
            check(lib().npp_mix_bn_bwd(C.cast(arr, C.c_void_p), k, wf.data_ptr(), _byref(dout), sums.data_ptr(), gw.data_ptr(), stream_ptr()),
                  "npp_mix_bn_bwd")
        g = None
        if need[0]:
            g = gw[:k] if ctx.wdtype == torch.float32 else gw[:k].to(ctx.wdtype)
        return (g, None, *dxs)


def mix_bn_sum(w, sides, training: bool):
    """sum_k w[k] * BN_k(x_k) for the candidates of a MixedOp given as BnSides (bn=None: the operand is added as is).  One fused
    launch when every BatchNorm side is affine-free with local batch statistics and the layout allows; otherwise each BatchNorm is
    applied on its own and the weighted sum follows (the path for SyncBatchNorm, eval mode and odd layouts)."""
    xs = []
    ok = MIX_FUSE and 1 <= len(sides) <= 8
    for sd in sides:
        x = to_nhwc(sd.x if sd.private else take(sd.x))
        sd.x = x
        xs.append(x)
    if ok:
        x0 = xs[0]
        # the kernels keep per-(operand, channel) coefficients in LDS next to a 36 KiB reduction image: 8 * k * C + 36 KiB <= 64 KiB
        ok = len(sides) * x0.shape[1] <= 3072
        for sd, x in zip(sides, xs):
            if x.shape != x0.shape or x.dtype != x0.dtype or not _fused_layout_ok(x):
                ok = False
            bn = sd.bn
            if bn is not None and not (training and bn.weight is None and bn.bias is None
                                       and (_local_batch_bn(bn, training) or _mix_sync_group(bn, training) is not None)):
                ok = False
        if ok:
            # SyncBatchNorm sides: all of one group, statistics through the peer-to-peer mailboxes (the sums of the backward pass
            # travel as one segment of (k + 1) * C values)
            grps = {id(_mix_sync_group(sd.bn, training)) for sd in sides if sd.bn is not None and _sync_group(sd.bn)[0] is not None}
            if len(grps) > 1:
                ok = False
            elif grps:
                from . import comm
                ok = P2P_DIRECT and MIX_SYNC and comm.p2p_can((len(sides) + 1) * x0.shape[1],
                                                               next(_sync_group(sd.bn)[0] for sd in sides
                                                                    if sd.bn is not None and _sync_group(sd.bn)[0] is not None))
    if ok:
        return _MixBnSum.apply(w, sides, *xs)
    ys = [None] * len(sides)
    bn_idx = [i for i, sd in enumerate(sides) if sd.bn is not None]
    for i, (sd, x) in enumerate(zip(sides, xs)):
        if sd.bn is None:
            ys[i] = x
        else:
            sd.private = True
    if BN_PAIRS and len(bn_idx) >= 2 and any(_sync_group(sides[i].bn)[0] is not None for i in bn_idx):
        # SyncBatchNorm: the candidates' BatchNorms as ONE autograd node -- their backward passes share one statistics exchange
        # instead of one each (the forward statistics already travel together: every candidate was pending before the first apply)
        outs = bn_add_multi([(sides[i], None, False, sides[i].bn.training, None) for i in bn_idx])
        for i, y in zip(bn_idx, outs):
            ys[i] = y
    else:
        for i in bn_idx:
            ys[i] = bn_add(sides[i], None, relu=False, training=sides[i].bn.training)
    return _WeightedSum.apply(w, *[take(y) for y in ys])


class _Interleave2(Function):
    """channel_shuffle(torch.cat([a, b], 1), groups=2)  (model_search_interact.py:22-36,71-72)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = to_nhwc(a), to_nhwc(b)
        n, c, h, w = a.shape
        out = new_nhwc(n, 2 * c, h, w, a.dtype, a.device)
        check(lib().npp_interleave2(_byref(a), _byref(b), _byref(out), 0, None, None, stream_ptr()), "npp_interleave2")
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = to_nhwc(dout)
        n, c2, h, w = dout.shape
        da = new_nhwc(n, c2 // 2, h, w, dout.dtype, dout.device)
        db = new_nhwc(n, c2 // 2, h, w, dout.dtype, dout.device)
        check(lib().npp_interleave2(None, None, _byref(dout), 1, _byref(da), _byref(db), stream_ptr()), "npp_interleave2(bwd)")
        return da, db


def interleave2(a, b):
    if b.dtype != a.dtype:
        b = cast(b, a.dtype)
    return _Interleave2.apply(take(a), take(b))


def scale_by(x, w_scalar):
    """w * x for a 0-d tensor w (the beta edge weights): a one-operand weighted sum."""
    return _WeightedSum.apply(w_scalar.reshape(1), take(x))
