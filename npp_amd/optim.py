"""Multi-tensor Adam on one HIP launch -- drop-in for `torch.optim.Adam(params, lr, betas, eps, weight_decay)` as the
reference builds it (augment_lip_sync.py:210-213: parameter groups with their own lr / weight decay, no amsgrad).

SURVEY §8f-2.  Parameters and gradients stay the tensors autograd and `state_dict()` know; what is new is a device-resident
job table (`NppAdamJob` per tensor + one (job, chunk) pair per block) so that the whole model -- 1756 tensors, 77 M
elements for NPPNet -- is ONE kernel at HBM speed instead of ~46 multi-tensor launches.  Per step the host only gathers the
gradient pointers (autograd hands out fresh gradient tensors in eager mode); the table is re-uploaded when one of them, or a
hyper-parameter, changed -- from pinned memory allocated beforehand, so the upload can sit inside a hipGraph capture.
There is no CPU path: parameters must live on the GPU (the product path fails loudly without the HIP library).
"""
from __future__ import annotations

import numpy as np
import torch

from ._lib import check, lib, stream_ptr

_JOB = np.dtype([("param", "<u8"), ("grad", "<u8"), ("exp_avg", "<u8"), ("exp_avg_sq", "<u8"), ("n", "<i8"),
                 ("lr", "<f4"), ("beta1", "<f4"), ("beta2", "<f4"), ("eps", "<f4"), ("weight_decay", "<f4"), ("_pad", "<i4")])
assert _JOB.itemsize == 64        # sizeof(NppAdamJob), include/npp_hip.h


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("FusedAdam: invalid hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._chunk = int(lib().npp_adam_chunk_elems())
        self._plist = None        # parameters that took part in the last table
        self._jobs = None         # numpy structured array (host image of the table)
        self._hyper = None
        self._pin = None          # (pinned jobs, pinned chunks)
        self._dev = None          # (device jobs, device chunks)
        self._step = None         # device int64
        self._captured_pin = None # index of the pinned image a hipGraph capture uploads from
        self._retired = []

    def _state_for(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["step"] = torch.zeros((), dtype=torch.float32)      # placeholder for state_dict compatibility
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _rebuild(self, plist, hyper, dev):
        n = len(plist)
        jobs = np.zeros(n, dtype=_JOB)
        sizes = np.empty(n, dtype=np.int64)
        for i, p in enumerate(plist):
            if p.dtype != torch.float32 or not p.is_contiguous() or not p.is_cuda:
                raise RuntimeError("FusedAdam: parameters must be contiguous float32 GPU tensors (no CPU path)")
            st = self._state_for(p)
            jobs["param"][i] = p.data_ptr()
            jobs["exp_avg"][i] = st["exp_avg"].data_ptr()
            jobs["exp_avg_sq"][i] = st["exp_avg_sq"].data_ptr()
            sizes[i] = p.numel()
        jobs["n"] = sizes
        h = np.asarray(hyper, dtype=np.float32)
        jobs["lr"], jobs["beta1"], jobs["beta2"], jobs["eps"], jobs["weight_decay"] = h[:, 0], h[:, 1], h[:, 2], h[:, 3], h[:, 4]
        nch = (sizes + self._chunk - 1) // self._chunk
        chunks = np.empty((int(nch.sum()), 2), dtype=np.int32)
        chunks[:, 0] = np.repeat(np.arange(n, dtype=np.int32), nch)
        starts = np.cumsum(nch) - nch
        chunks[:, 1] = np.arange(int(nch.sum()), dtype=np.int64) - np.repeat(starts, nch)
        self._jobs, self._plist, self._hyper = jobs, plist, hyper
        # two pinned images of the job table, used alternately: the host never waits for the upload of the previous step
        pjs = [torch.empty(n * _JOB.itemsize, dtype=torch.uint8).pin_memory() for _ in range(2)]
        pc = torch.from_numpy(chunks.reshape(-1).copy()).pin_memory()
        if self._pin is not None and self._captured_pin is not None:
            self._retired.append((self._pin, self._dev))      # a captured graph still uploads from / into the old tables
        self._pin = (pjs, pc)
        self._captured_pin = None
        self._pin_ev = [None, None]
        self._pin_i = 0
        self._dev = (torch.empty(pjs[0].numel(), dtype=torch.uint8, device=dev), torch.empty(pc.numel(), dtype=torch.int32, device=dev))
        self._dev[1].copy_(pc, non_blocking=True)
        self._nchunks = int(chunks.shape[0])
        if self._step is None:
            self._step = torch.full((1,), int(getattr(self, "_loaded_step", 0)), dtype=torch.int64, device=dev)
        self._grad_ptrs = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        plist, hyper = [], []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            h = (float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]))
            for p in group["params"]:
                if p.grad is not None:
                    plist.append(p)
                    hyper.append(h)
        if not plist:
            return loss
        same = self._plist is not None and len(plist) == len(self._plist) and all(a is b for a, b in zip(plist, self._plist))
        if not same:
            if self._plist is not None and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FusedAdam: the parameter set changed inside a hipGraph capture "
                                   "(run one eager step with the same set first: pinned tables are allocated there)")
            self._rebuild(plist, hyper, plist[0].device)
        elif hyper != self._hyper:
            self._apply_hyper(hyper)
        for p in plist:
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                p.grad = g.float().contiguous()
        gp = np.fromiter((p.grad.data_ptr() for p in plist), dtype=np.uint64, count=len(plist))
        capturing = torch.cuda.is_current_stream_capturing()
        # under capture the upload is always recorded: a replay re-reads the pinned image, which is how refresh_hyper()
        # reaches a captured step
        cap = self._captured_pin
        if capturing or cap is not None or self._grad_ptrs is None or not np.array_equal(gp, self._grad_ptrs):
            # once a step has been captured its pinned image belongs to the graph (every replay uploads it, so the device
            # table must also be re-uploaded by every eager step): eager steps keep to the other image
            if cap is not None and not capturing:
                i = 1 - cap
            else:
                i = self._pin_i
                self._pin_i ^= 1
            if self._pin_ev[i] is not None and not capturing:
                self._pin_ev[i].synchronize()         # the upload issued two steps ago has read this pinned image
            self._jobs["grad"] = gp
            self._pin[0][i].numpy()[:] = self._jobs.view(np.uint8)
            self._dev[0].copy_(self._pin[0][i], non_blocking=True)
            if not capturing:
                self._pin_ev[i] = torch.cuda.Event()
                self._pin_ev[i].record()
            self._grad_ptrs = gp
            if capturing:
                self._captured_pin = i
        check(lib().npp_adam_step(self._dev[0].data_ptr(), self._dev[1].data_ptr(), self._nchunks, self._step.data_ptr(),
                                  stream_ptr()), "npp_adam_step")
        torch.autograd.graph.increment_version(plist)   # in-place update behind autograd's back: derived caches must see it
        return loss

    def _apply_hyper(self, hyper):
        """New lr / betas / eps / weight decay for the SAME parameter set: rewrite the columns of the job table in place
        (no reallocation: a captured hipGraph keeps uploading from the same pinned image into the same device table)."""
        h = np.asarray(hyper, dtype=np.float32)
        j = self._jobs
        j["lr"], j["beta1"], j["beta2"], j["eps"], j["weight_decay"] = h[:, 0], h[:, 1], h[:, 2], h[:, 3], h[:, 4]
        self._hyper = hyper
        self._grad_ptrs = None                 # the next eager step uploads
        cap = self._captured_pin
        if cap is not None and not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize()           # no replay may be reading the pinned image while it is rewritten
            keep = self._pin[0][cap].numpy().view(_JOB)["grad"].copy()      # the graph's own gradient addresses
            img = j.copy()
            img["grad"] = keep
            self._pin[0][cap].numpy()[:] = img.view(np.uint8)

    def refresh_hyper(self):
        """Carry changed param_group hyper-parameters (a learning-rate schedule, augment_lip_sync.py:213,249) into a step
        that was captured in a hipGraph: the captured upload copies the pinned job table at every replay, so rewriting the
        lr / betas / eps / weight-decay columns of that pinned image is enough.  The parameter set must be unchanged."""
        if self._plist is None:
            return
        hyper = []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            h = (float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]))
            hyper.extend(h for p in group["params"] if p.grad is not None)
        if len(hyper) != len(self._plist):
            raise RuntimeError("FusedAdam.refresh_hyper: the set of parameters with gradients changed since the last step")
        if hyper != self._hyper:
            self._apply_hyper(hyper)

    def device_step_count(self) -> int:
        return int(self._step.item()) if self._step is not None else 0

    # -- checkpointing (augment_lip_sync.py:235,268-278 / search_lip_sync.py:309 save and resume the optimizer) ---------------
    def state_dict(self):
        """torch.optim.Adam's layout: the step count lives on the device (`_step`, one counter for all parameters); it is
        written into every `state[p]['step']` here, so that a checkpoint resumes with the right bias correction -- in this
        class and in `torch.optim.Adam`."""
        n = self.device_step_count()
        for st in self.state.values():
            if "exp_avg" in st:
                st["step"] = torch.tensor(float(n), dtype=torch.float32)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Restores moments AND the step count; the job table is rebuilt at the next step (it pointed at the old moment
        tensors).  A step captured in a hipGraph before this call keeps the old tensors: capture after loading."""
        super().load_state_dict(state_dict)
        steps = [float(st["step"]) for st in self.state.values() if "step" in st]
        dev = None
        for st in self.state.values():
            if "exp_avg" in st:
                dev = st["exp_avg"].device
                st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32).cpu() if "step" in st else \
                    torch.zeros((), dtype=torch.float32)
        n = int(round(max(steps))) if steps else 0
        if self._step is not None:
            self._step.fill_(n)
        elif dev is not None and dev.type == "cuda":
            self._step = torch.full((1,), n, dtype=torch.int64, device=dev)
        else:
            self._loaded_step = n          # parameters not on the GPU yet: applied when the table is first built
        self._plist = None
        self._jobs = None
        self._grad_ptrs = None
