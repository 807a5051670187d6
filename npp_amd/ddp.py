"""Bucketed gradient all-reduce for one-process-per-GPU data parallelism (RCCL over xGMI).

Replaces what `nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)` does for the
reference (augment_lip_sync.py:206-208) with a reducer shaped for this network and
this node:

  * parameters that never receive a gradient are excluded STATICALLY (`unused_parameter_names`: the 116
    tensors of SE_Block.bn when stride == 1, operations.py:117,126-129), so no per-step graph walk
    (`find_unused_parameters`) is needed;
  * buckets are filled in reverse registration order (~ the order gradients appear in backward) and sized
    for xGMI rings (default 32 MiB: 8 GPUs x 7 links, per-link bound);
  * a bucket's all-reduce is issued from the autograd thread the moment its last gradient has been
    accumulated, on a SIDE HIP stream fenced by an event on the compute stream, so the collective overlaps
    the rest of backward; `finish()` (before optimizer.step) joins the streams and writes the averaged
    gradients back.

`torch.distributed` is the transport ("nccl" == RCCL on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Set

import torch
import torch.distributed as dist


def unused_parameter_names(model: torch.nn.Module) -> Set[str]:
    """Names of parameters that cannot receive a gradient in `model`'s forward."""
    from .operations import SE_Block
    names = set()
    for mname, m in model.named_modules():
        if isinstance(m, SE_Block) and m.stride == 1:
            for pname, _ in m.bn.named_parameters():
                names.add(f"{mname}.bn.{pname}" if mname else f"bn.{pname}")
    return names


class _Bucket:
    __slots__ = ("params", "offsets", "flat", "pending", "work", "numel", "streams")

    def __init__(self, params, device, dtype):
        self.params = params
        self.offsets = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += p.numel()
        self.numel = off
        self.flat = torch.zeros(off, dtype=dtype, device=device)
        self.pending = len(params)
        self.work = None
        self.streams = {}      # compute streams the gradients of this bucket were accumulated on (the network runs
                               # its two task branches, and therefore their backward, on two streams)


class GradReducer:
    def __init__(self, module: torch.nn.Module, process_group=None, bucket_mb: float = 32.0,
                 skip: Optional[Iterable[str]] = None, broadcast_parameters: bool = True, always_reduce: bool = False,
                 overlap: bool = True):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GradReducer needs an initialised torch.distributed process group")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.always = always_reduce      # run the collectives even on a 1-rank group (single-GPU exercise of the path)
        # overlap=False: the bucket all-reduces are issued by finish() on the caller's stream instead of from the
        # autograd hooks on a side stream.  Meant for hipGraph capture, where every collective of the step then sits on
        # the capture's origin stream (the unoverlapped 308 MB all-reduce is ~2 ms on an 8-GPU xGMI node).
        self.overlap = overlap
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
        cur, cur_bytes = [], 0
        for n, p in reversed(named):
            nb = p.numel() * p.element_size()
            if cur and (cur_bytes + nb > cap or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self.buckets.append(_Bucket(cur, cur[0].device, cur[0].dtype))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self.buckets.append(_Bucket(cur, cur[0].device, cur[0].dtype))
        self._where = {}
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for pi, p in enumerate(b.params):
                self._where[p] = (bi, pi)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        dev = self.buckets[0].flat.device if self.buckets else torch.device("cpu")
        self._cuda = dev.type == "cuda"
        self._side = torch.cuda.Stream(device=dev) if self._cuda else None
        self._launched = 0

    # called by autograd (possibly on its own thread) right after p.grad has been accumulated
    def _on_grad(self, p):
        bi, pi = self._where[p]
        b = self.buckets[bi]
        if self._cuda:
            st = torch.cuda.current_stream()
            b.streams[st.cuda_stream] = st
        b.pending -= 1
        if b.pending == 0 and self.overlap:
            self._launch(b)

    def _launch(self, b: _Bucket):
        if self._cuda:
            # fence the side stream behind every stream that accumulated one of this bucket's gradients
            for st in b.streams.values():
                ev = torch.cuda.Event()
                ev.record(st)
                self._side.wait_event(ev)
            b.streams = {}
            ctx = torch.cuda.stream(self._side)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx, torch.no_grad():
            views = [b.flat[o:o + p.numel()].view_as(p) for o, p in zip(b.offsets, b.params)]
            torch._foreach_copy_(views, [p.grad for p in b.params])
            if self.world > 1 or self.always:
                b.work = dist.all_reduce(b.flat, group=self.group, async_op=True)
        self._launched += 1

    def finish(self):
        """Join the collectives and install the averaged gradients.  Call once per step after backward()."""
        scale = 1.0 / self.world
        for b in self.buckets:
            if b.pending != 0:
                missing = [i for i, p in enumerate(b.params) if p.grad is None]
                raise RuntimeError(f"GradReducer: a bucket never completed ({b.pending} gradients missing, "
                                   f"{len(missing)} parameters without grad): add them to `skip`")
            if b.work is not None:
                b.work.wait()
                b.work = None
        if not self.overlap:
            with torch.no_grad():
                for b in self.buckets:
                    views = [b.flat[o:o + p.numel()].view_as(p) for o, p in zip(b.offsets, b.params)]
                    torch._foreach_copy_(views, [p.grad for p in b.params])
                    if self.world > 1 or self.always:
                        dist.all_reduce(b.flat, group=self.group)
                    b.streams = {}
        elif self._cuda:
            torch.cuda.current_stream().wait_stream(self._side)
        with torch.no_grad():
            for b in self.buckets:
                if self.world > 1 or self.always:
                    b.flat.mul_(scale)
                    views = [b.flat[o:o + p.numel()].view_as(p) for o, p in zip(b.offsets, b.params)]
                    torch._foreach_copy_([p.grad for p in b.params], views)
                b.pending = len(b.params)
        self._launched = 0

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
