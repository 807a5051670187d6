"""Bucketed gradient all-reduce for one-process-per-GPU data parallelism (RCCL over xGMI).

Replaces what `nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)` does for the
reference (augment_lip_sync.py:206-208) with a reducer shaped for this network and
this node:

  * parameters that never receive a gradient are excluded STATICALLY (`unused_parameter_names`: the 116
    tensors of SE_Block.bn when stride == 1, operations.py:117,126-129), so no per-step graph walk
    (`find_unused_parameters`) is needed;
  * buckets are filled in reverse registration order (~ the order gradients appear in backward) and sized
    for xGMI rings (default 16 MiB: 8 GPUs x 7 links, per-link bound);
  * every bucket is ONE pre-flattened f32 buffer and the gradients are VIEWS of it: the weight-gradient /
    BatchNorm / bias kernels of this package write their results straight into the parameter's slot
    (`_ops.grad_out`), autograd installs that view as `p.grad`, the collective reduces the buffer in place
    and the optimizer reads the averaged views -- no flatten / un-flatten passes over the 308 MB of
    gradients (a gradient that did not land in its slot, e.g. one made by plain autograd math, is copied in
    when its bucket closes);
  * a bucket's all-reduce is issued from the autograd thread the moment its last gradient has been
    accumulated -- eagerly on a SIDE HIP stream fenced by events on the compute streams; inside a hipGraph
    capture on the capture's ORIGIN stream (the one stream RCCL may be captured on with this ROCm, see
    `_ops.hub_all_reduce`), which the two task-branch streams only feed with one-way events -- so the
    collective overlaps the rest of backward; `finish()` (before optimizer.step) joins and averages.

  * round 3, `overlap="tail"` (what bench.py's N > 1 runs use as the overlapped form): the step keeps its batched weight-gradient
    tail (TrainStep defers every weight gradient into a few launches after backward), so there is nothing to overlap with DURING
    backward -- the overlap moves into the tail: buckets are built per kind, the dense KxK conv weights (4/5 of the bytes) become
    final after the tail's first group (KxK launches + unpack) and are reduced while the second group (1x1, depthwise, SE) is
    computed on the other stream; `finish()` reduces the small rest.  One rank: 51.4 ms per step, the same as reducing everything
    after backward (51.5) and 9.5 ms less than the hook form (60.9), which gives the batching up.

`torch.distributed` is the transport ("nccl" == RCCL on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Set

import torch
import torch.distributed as dist

from . import _ops as K
from . import comm as _comm


def unused_parameter_names(model: torch.nn.Module) -> Set[str]:
    """Names of parameters that cannot receive a gradient in `model`'s forward."""
    from .operations import SE_Block
    names = set()
    for mname, m in model.named_modules():
        if isinstance(m, SE_Block) and m.stride == 1:
            for pname, _ in m.bn.named_parameters():
                names.add(f"{mname}.bn.{pname}" if mname else f"bn.{pname}")
    return names


def _is_kxk_weight(p) -> bool:
    """A dense KxK conv weight: its gradient is accumulated in the packed layout and becomes final in the FIRST group of the
    batched weight-gradient tail (KxK launches + unpack), see GradReducer(overlap="tail")."""
    return p.dim() == 4 and p.shape[2] * p.shape[3] > 1 and p.shape[1] > 1


class _Bucket:
    __slots__ = ("params", "offsets", "flat", "pending", "numel", "streams", "taken", "launched", "kind")

    def __init__(self, params, device, dtype, kind="O"):
        self.params = params
        self.kind = kind
        self.offsets = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4        # 16-byte aligned slots (the kernels store vectors)
        self.numel = off
        self.flat = torch.zeros(off, dtype=dtype, device=device)
        self.pending = len(params)
        self.launched = False
        self.taken = [False] * len(params)
        self.streams = {}      # compute streams the gradients of this bucket were accumulated on (the network runs
                               # its two task branches, and therefore their backward, on two streams)

    def view(self, i):
        p = self.params[i]
        o = self.offsets[i]
        return self.flat[o:o + p.numel()].view(p.shape)


class GradReducer:
    def __init__(self, module: torch.nn.Module, process_group=None, bucket_mb: Optional[float] = None,
                 skip: Optional[Iterable[str]] = None, broadcast_parameters: bool = True, always_reduce: bool = False,
                 overlap: bool = True):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GradReducer needs an initialised torch.distributed process group")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.always = always_reduce      # run the collectives even on a 1-rank group (single-GPU exercise of the path)
        # overlap: True / "hooks" -- a bucket is reduced the moment its last gradient lands in backward (the step then cannot defer and
        #            batch its weight gradients: every one of them is a launch of its own on the branch streams);
        #          "tail"  -- the step keeps its batched weight-gradient tail and the reducer overlaps with THAT: the buckets are
        #            built per kind, the KxK conv weights (4/5 of the bytes) become final in the tail's first group and are reduced
        #            while the second group (1x1, depthwise, SE) is computed (train_step.TrainStep drives it);
        #          False   -- every bucket is reduced by finish() on the caller's stream, after backward (NPP_DDP_OVERLAP=0)
        env = os.environ.get("NPP_DDP_OVERLAP")
        if env is not None:
            overlap = {"0": False, "1": True, "hooks": True, "tail": "tail"}.get(env, overlap)
        self.tail = overlap == "tail"
        self.overlap = bool(overlap) and not self.tail      # (what TrainStep asks: do hooks read gradients during backward?)
        if bucket_mb is None:
            bucket_mb = float(os.environ.get("NPP_DDP_BUCKET_MB", "16"))
        skip = set(skip or ())
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad and n not in skip]
        self.skipped = sorted(skip)
        if broadcast_parameters and (self.world > 1 or always_reduce):
            # DDP construction broadcasts rank 0's parameters and buffers
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t, 0, group=process_group)
        cap = int(bucket_mb * 1024 * 1024)
        self.buckets: List[_Bucket] = []
        groups = [("O", list(reversed(named)))]
        if self.tail:      # one run of buckets per kind: a bucket never mixes gradients that become final in different groups of the tail
            groups = [("K", [(n, p) for n, p in reversed(named) if _is_kxk_weight(p)]),
                      ("O", [(n, p) for n, p in reversed(named) if not _is_kxk_weight(p)])]
        for kind, plist in groups:
            cur, cur_bytes = [], 0
            for n, p in plist:
                nb = p.numel() * p.element_size()
                if cur and (cur_bytes + nb > cap or p.dtype != cur[0].dtype or p.device != cur[0].device):
                    self.buckets.append(_Bucket(cur, cur[0].device, cur[0].dtype, kind))
                    cur, cur_bytes = [], 0
                cur.append(p)
                cur_bytes += nb
            if cur:
                self.buckets.append(_Bucket(cur, cur[0].device, cur[0].dtype, kind))
        self._where = {}
        self._hooks = []
        self._slot_keys = []
        for bi, b in enumerate(self.buckets):
            for pi, p in enumerate(b.params):
                self._where[p] = (bi, pi)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
                if p.is_cuda:
                    K.register_grad_slot(p, self._slot)
                    self._slot_keys.append(p.data_ptr())
        dev = self.buckets[0].flat.device if self.buckets else torch.device("cpu")
        self._cuda = dev.type == "cuda"
        self._side = torch.cuda.Stream(device=dev) if self._cuda else None
        self._armed = False
        backend = dist.get_backend(process_group)
        # transport: the library's own RCCL communicator (npp_amd.comm, NPP_COMM=npp) or torch.distributed's
        self._npp = False
        if self._cuda and backend == "nccl" and (_comm.wanted() or _comm.active()):
            _comm.enable(process_group)
            self._npp = _comm._state["group"] is process_group
        self._avg = backend == "nccl" or self._npp   # RCCL averages in the collective; gloo has no AVG: sum, then one scale pass

    # -- gradient slots: the kernels write a parameter's gradient straight into its place in the bucket ---------------------
    def begin_step(self):
        """Call after zero_grad(set_to_none=True), before backward: zeroes the buckets (the 1x1 weight-gradient kernels
        accumulate with atomics) and lets the backward kernels write into them.  Without it every gradient is copied into its
        slot when its bucket closes (still correct, one more pass)."""
        if not self._cuda:
            return
        for b in self.buckets:
            b.flat.zero_()
            b.taken = [False] * len(b.params)
        self._armed = True

    def _slot(self, p, zero):
        """`_ops.grad_out` provider: the view of `p`'s bucket slot, once per step (a second gradient of the same parameter
        must be ADDED by autograd, not overwrite the first)."""
        if not self._armed:
            return None
        bi, pi = self._where[p]
        b = self.buckets[bi]
        if b.taken[pi] or b.launched or p.grad is not None:
            # p.grad is still set (zero_grad(set_to_none=False), or a second backward of an accumulation window): it may BE this
            # very slot, and autograd will add the new gradient to it -- the kernel must write somewhere else
            return None
        b.taken[pi] = True
        return b.view(pi)

    # called by autograd (possibly on its own thread) right after p.grad has been accumulated
    def _on_grad(self, p):
        bi, pi = self._where[p]
        b = self.buckets[bi]
        if self._cuda:
            st = torch.cuda.current_stream()
            b.streams[st.cuda_stream] = st
        b.pending -= 1
        if b.pending < 0:
            raise RuntimeError("GradReducer: a parameter's gradient arrived twice before finish() -- call finish() after every "
                               "backward (gradient accumulation over several backwards is not supported by this reducer)")
        if b.pending == 0 and self.overlap:
            self._launch(b)

    def _target_stream(self):
        """Stream the collectives go to, or None for the current one.  A step that is (to be) replayed as a hipGraph keeps
        every collective on the capture's origin stream (= the stream Network.forward was called on, `_ops._hub_stream`)."""
        if not self._cuda:
            return None
        if K.GRAPH_TOPOLOGY or torch.cuda.is_current_stream_capturing():
            return K._hub_stream
        return self._side

    def _launch(self, b: _Bucket, target="auto"):
        import contextlib
        tgt = self._target_stream() if target == "auto" else target
        if self._cuda:
            # fence the collective's stream behind every stream that accumulated one of this bucket's gradients
            dst_stream = tgt if tgt is not None else torch.cuda.current_stream()
            for st in b.streams.values():
                if st.cuda_stream != dst_stream.cuda_stream:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    dst_stream.wait_event(ev)
        b.streams = {}
        ctx = torch.cuda.stream(tgt) if tgt is not None else contextlib.nullcontext()
        with ctx, torch.no_grad():
            src, dst = [], []
            for i, p in enumerate(b.params):
                g = p.grad
                v = b.view(i)
                if g.data_ptr() != v.data_ptr() or g.dtype != v.dtype or not g.is_contiguous():
                    src.append(g)
                    dst.append(v)
                    p.grad = v
            if src:
                torch._foreach_copy_(dst, src)
            if self.world > 1 or self.always:
                # the stream-synchronous form: the collective is ordered after, and joined back into, THIS stream (inside a
                # capture that keeps RCCL's internal stream in a two-way relation with the capture's origin stream only)
                if self._npp:
                    _comm.all_reduce_bucket(b.flat, average=True)
                else:
                    dist.all_reduce(b.flat, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.group)
        b.launched = True

    def launch_kind(self, kind, target="auto"):
        """overlap="tail": reduce every bucket of `kind` whose gradients are complete NOW (the caller has just run the group of the
        weight-gradient tail that finishes them) on the collectives' stream, fenced behind the caller's stream."""
        tgt = self._target_stream() if target == "auto" else target
        if self._cuda and tgt is not None and tgt.cuda_stream != torch.cuda.current_stream().cuda_stream:
            tgt.wait_stream(torch.cuda.current_stream())
        for b in self.buckets:
            if b.kind == kind and not b.launched:
                if b.pending != 0:
                    raise RuntimeError(f"GradReducer: a bucket of kind {kind} is incomplete ({b.pending} gradients missing)")
                self._launch(b, target=tgt)

    def finish(self):
        """Join the collectives; afterwards every p.grad is the averaged view of its bucket.  Call once per step after
        backward() and before optimizer.step()."""
        for b in self.buckets:
            if b.pending != 0:
                missing = [i for i, p in enumerate(b.params) if p.grad is None]
                raise RuntimeError(f"GradReducer: a bucket never completed ({b.pending} gradients missing, "
                                   f"{len(missing)} parameters without grad): add them to `skip`")
        for b in self.buckets:
            if not b.launched:
                self._launch(b, target=None)          # overlap=False: here, on the caller's stream
        if self._cuda:
            cur, tgt = torch.cuda.current_stream(), self._target_stream()
            if tgt is not None and tgt.cuda_stream != cur.cuda_stream:
                cur.wait_stream(tgt)
        if (self.world > 1 or self.always) and not self._avg:
            with torch.no_grad():
                torch._foreach_mul_([b.flat for b in self.buckets], 1.0 / self.world)
        self.reset()

    def reset(self):
        """Forget the per-step state (also after a step that was abandoned half way, e.g. a failed hipGraph capture)."""
        for b in self.buckets:
            b.pending = len(b.params)
            b.streams = {}
            b.launched = False
            b.taken = [False] * len(b.params)
        self._armed = False

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for k in self._slot_keys:
            K.unregister_grad_slot(k)
        self._slot_keys = []
