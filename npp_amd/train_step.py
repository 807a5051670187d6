"""One optimisation step of the reference's training loop as a callable, replayed as a hipGraph.

Counterpart of the loop body of `train()` (core/function.py:72-107):

    output_pose, output_par = model(images)
    loss = (criterion_par(output_par, labels_par).unsqueeze(0)
            + criterion_pose(output_pose, labels_pose, target_weight=pose_weight).unsqueeze(0)).mean()
    model.zero_grad(); loss.backward(); optimizer.step()

A step of NPPNet is ~5 000 kernel launches; issued eagerly from Python the host is the bottleneck (~115 img/s at batch 16 on
an MI355X, where the GPU alone does ~290).  The step is static -- fixed genotype, fixed shapes, no host decision between
launches -- so `TrainStep` runs the first `warmup` calls eagerly (they are real steps on real batches; they also size
every pool), captures the next call into ONE hipGraph (forward, both criteria, backward, the gradient all-reduce and the
SyncBatchNorm exchanges when a process group is active, and the optimizer) and from then on copies the batch into the
graph's static input buffers and replays it.  Learning-rate changes (`MultiStepLR`, augment_lip_sync.py:213,249) reach the
replayed graph through `FusedAdam.refresh_hyper()`: the captured table upload re-reads pinned memory at every replay.

A call with different input shapes (the last, short batch of an epoch) runs eagerly.  If capture fails the step stays
eager and says so once on stderr; `NPP_TRAIN_GRAPH=0` forces that.  The returned loss is a 0-dim device tensor (a static
buffer of the graph: `.item()` / `reduce_tensor` it before the next call, as the reference's loop does).
"""
from __future__ import annotations

import os
import sys
import time
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _ops as K


class TrainStep:
    def __init__(self, model, criterion_pose, criterion_par, optimizer, reducer=None, graph: Optional[bool] = None,
                 warmup: int = 2):
        self.model, self.criterion_pose, self.criterion_par = model, criterion_pose, criterion_par
        for m in (model, getattr(model, "module", None)):
            if m is not None:
                m._auto_graph_off = True      # this step captures the whole loop body itself (npp_amd/auto_graph.py stays out)
        self.optimizer, self.reducer = optimizer, reducer
        if graph is None:
            graph = os.environ.get("NPP_TRAIN_GRAPH", "1") != "0"
        self.use_graph = bool(graph)
        if self.use_graph and dist.is_available() and dist.is_initialized() and dist.get_backend() != "nccl":
            # only RCCL ("nccl") collectives can sit inside a hipGraph; a gloo all-reduce invalidates the capture (and its
            # own device streams with it: the eager fallback then dies inside gloo)
            self.use_graph = False
        K.GRAPH_TOPOLOGY = self.use_graph      # warm-up steps must already run the topology that gets captured
        self.warmup = max(int(warmup), 1)      # at least one eager step: pools, pinned optimizer tables, packed weights
        self.calls = 0
        self.graph = None
        self._static_in: Optional[List[torch.Tensor]] = None
        self._static_loss = None
        self._sig = None
        self._hyper = None
        self._side = None

    # -- the step itself (what gets captured) ---------------------------------------------------------------------
    def _loss(self, images, labels_par, labels_pose, pose_weight):
        output_pose, output_par = self.model(images)
        losses_par = self.criterion_par(output_par, labels_par).unsqueeze(0)
        if pose_weight is not None:
            losses_pose = self.criterion_pose(output_pose, labels_pose, target_weight=pose_weight).unsqueeze(0)
        else:
            losses_pose = self.criterion_pose(output_pose, labels_pose).unsqueeze(0)
        return (losses_par + losses_pose).mean()

    def _eager(self, images, labels_par, labels_pose, pose_weight):
        K.stamp("step begin")
        loss = self._loss(images, labels_par, labels_pose, pose_weight)
        K.stamp("loss fwd done")
        self.optimizer.zero_grad(set_to_none=True)
        if self.reducer is not None:
            self.reducer.begin_step()      # the backward kernels write the gradients straight into the reducer's buckets
        # no gradient exists yet (set_to_none) and nothing reads one before the optimizer: the KxK weight-gradient unpacks of
        # this backward are collected and run as one launch (a reducer's bucket hooks read gradients during backward: not then)
        # (a reducer that launches its buckets from finish() -- overlap=False, NPP_DDP_OVERLAP=0 -- reads no gradient before that
        # either: the deferred launches then write straight into its bucket slots and finish() follows them)
        hooks_read = self.reducer is not None and getattr(self.reducer, "overlap", True)
        K.DEFER_UNPACK = not hooks_read and os.environ.get("NPP_DEFER_UNPACK", "1") != "0"
        K.DEFER_WGRAD_MAX_PIX = int(os.environ.get("NPP_DEFER_WGRAD_MAX_PIX", "150000")) if K.DEFER_UNPACK else 0
        try:
            loss.backward()
        finally:
            K.DEFER_UNPACK = False
            K.DEFER_WGRAD_MAX_PIX = 0
        K.stamp("backward done")
        if self.reducer is not None and getattr(self.reducer, "tail", False) and K.DEFER_TAIL_OK:
            self._tail_with_reducer()
        else:
            self._tail()
            K.stamp("weight gradients done")
            if self.reducer is not None:
                self.reducer.finish()
        self.optimizer.step()
        K.stamp("optimizer done")
        return loss

    def _tail(self):
        """The batched weight-gradient launches of the step.  NPP_TAIL_SPLIT=1: the KxK group on the caller's stream and the rest (1x1,
        depthwise, SE, their unpacks) on the parsing branch's stream at the same time -- each batched launch ends in a tail of a few
        long workgroups during which most CUs idle; two launches in flight fill each other's tails."""
        cur = torch.cuda.current_stream() if torch.cuda.is_available() else None
        if cur is None or os.environ.get("NPP_TAIL_SPLIT", "0") != "1":
            K.flush_wgrads()
            K.flush_unpacks()
            return
        from .model_augment import _side_stream
        side = _side_stream(cur.device, 0)
        side.wait_stream(cur)
        K.flush_wgrads(group="K")
        K.flush_unpacks(group="K")
        with torch.cuda.stream(side):
            K.flush_wgrads()
            K.flush_unpacks()
        cur.wait_stream(side)

    def _tail_with_reducer(self):
        """GradReducer(overlap="tail"): the batched weight-gradient tail in two groups with the all-reduce of the first group's
        buckets (the KxK conv weights, 4/5 of the gradient bytes) running under the second group's kernels.  The collectives stay on
        the stream the reducer names (inside a capture: the capture's origin, the only stream RCCL may be captured on here); the
        second group's launches go to the parsing branch's stream, which forks from and joins the origin."""
        cur = torch.cuda.current_stream()
        K.flush_wgrads(group="K")
        K.flush_unpacks(group="K")               # every KxK weight gradient is final
        tgt = self.reducer._target_stream()
        same = tgt is None or tgt.cuda_stream == cur.cuda_stream
        side = None
        if same and cur.device.type == "cuda":
            # the collectives share the caller's stream (a captured / to-be-captured step): the second group must run elsewhere
            from .model_augment import _side_stream
            side = _side_stream(cur.device, 0)
            side.wait_stream(cur)
        self.reducer.launch_kind("K")
        if side is not None:
            with torch.cuda.stream(side):
                K.flush_wgrads()
                K.flush_unpacks()                # (1x1 weights with padded packed rows)
            cur.wait_stream(side)
        else:
            K.flush_wgrads()
            K.flush_unpacks()
        K.stamp("weight gradients done")
        self.reducer.finish()

    # -- static buffers -----------------------------------------------------------------------------------------------
    @staticmethod
    def _flatten(images, labels_par, labels_pose, pose_weight):
        lp = list(labels_pose) if isinstance(labels_pose, (list, tuple)) else [labels_pose]
        flat = [images] + list(labels_par) + lp + ([pose_weight] if pose_weight is not None else [])
        layout = (len(labels_par), len(lp), isinstance(labels_pose, (list, tuple)), pose_weight is not None)
        return flat, layout

    @staticmethod
    def _unflatten(flat: Sequence[torch.Tensor], layout):
        npar, npose, pose_is_list, has_w = layout
        images = flat[0]
        labels_par = list(flat[1:1 + npar])
        lp = list(flat[1 + npar:1 + npar + npose])
        labels_pose = lp if pose_is_list else lp[0]
        pose_weight = flat[1 + npar + npose] if has_w else None
        return images, labels_par, labels_pose, pose_weight

    @staticmethod
    def _signature(flat, layout):
        return layout, tuple((tuple(t.shape), t.dtype, t.device) for t in flat)

    def _hyper_now(self):
        return [(float(g["lr"]), tuple(float(b) for b in g["betas"]), float(g["eps"]), float(g["weight_decay"]))
                for g in self.optimizer.param_groups]

    def _barrier(self):
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # -- capture ------------------------------------------------------------------------------------------------------
    def _capture(self, flat, layout):
        """Capture one step on static copies of `flat`.  Nothing executes during capture: the caller replays."""
        self._static_in = [t.clone() for t in flat]
        args = self._unflatten(self._static_in, layout)
        collectives = dist.is_available() and dist.is_initialized()
        if collectives:
            dist.barrier()
        torch.cuda.synchronize()
        if collectives:
            # the ProcessGroup watchdog polls the events of the eager collectives issued so far from its own thread; let it
            # retire them, and capture in thread-local mode so its polling can never be an illegal call inside the capture
            time.sleep(1.0)
        K.reset_pools()
        self.optimizer.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        origin = torch.cuda.current_stream()
        import gc
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()       # no cyclic collection inside the capture: destroying another hipGraph, its pool or a pinned block there is
        try:               # an illegal call (and, thrown from a destructor, an abort)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._static_loss = self._eager(*args)
                if os.environ.get("NPP_TEST_FAIL_CAPTURE"):      # test hook: an illegal call invalidates the capture
                    torch.cuda.current_stream().synchronize()
        except BaseException:
            # torch.cuda.graph.__exit__ raises from capture_end() BEFORE it restores the stream: the current stream would
            # stay the (invalidated) capture stream and every later launch would fail (tools/capture_recover.py)
            torch.cuda.set_stream(origin)
            raise
        finally:
            K.reset_pools()        # chunks handed out during capture belong to the graph's private pool
            if gc_was:
                gc.enable()
        self.graph = g
        self._sig = self._signature(flat, layout)
        self._hyper = self._hyper_now()

    def __call__(self, images, labels_par, labels_pose, pose_weight=None):
        self.calls += 1
        flat, layout = self._flatten(images, labels_par, labels_pose, pose_weight)
        if not self.use_graph:
            loss = self._eager(images, labels_par, labels_pose, pose_weight)
            self._p2p_health()
            return loss
        if self.graph is None:
            if self.calls <= self.warmup:
                # eager warm-up on a side stream (allocations of these steps must not land in the capture's pool)
                if self._side is None:
                    self._side = torch.cuda.Stream()
                self._side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._side):
                    loss = self._eager(images, labels_par, labels_pose, pose_weight)
                torch.cuda.current_stream().wait_stream(self._side)
                return loss
            self._check_p2p_transport()
            err = None
            try:
                self._capture(flat, layout)
            except Exception as exc:      # noqa: BLE001 -- any capture failure: stay eager
                err = exc
            # Every rank must take the SAME path from here on: a rank that fell back to eager runs the single-stream issue
            # order of the SyncBatchNorm exchanges while a replaying rank runs the lockstep order -- mismatched collectives,
            # i.e. a hang.  So the ranks agree (MIN over a "captured OK" flag, outside any capture) and all fall back if one
            # of them failed.
            if err is not None:
                self._abandon_capture(True)
            ok = self._all_ranks_ok(err is None)
            if not ok:
                if err is None:
                    err = RuntimeError("another rank failed to capture")
                    self._abandon_capture(False)
                sys.stderr.write(f"[npp_amd.TrainStep] hipGraph capture failed ({type(err).__name__}: {err}); running eager\n")
                return self._eager(images, labels_par, labels_pose, pose_weight)
            self.graph.replay()            # the captured step has not run yet: this executes it on this batch
            K.note_training_step()
            return self._static_loss
        if self._signature(flat, layout) != self._sig:
            return self._eager(images, labels_par, labels_pose, pose_weight)       # e.g. the short last batch
        for dst, src in zip(self._static_in, flat):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        hyper = self._hyper_now()
        if hyper != self._hyper:
            refresh = getattr(self.optimizer, "refresh_hyper", None)
            if refresh is None:
                raise RuntimeError("TrainStep: optimizer hyper-parameters changed after capture and the optimizer has no "
                                   "refresh_hyper() (use npp_amd.optim.FusedAdam)")
            refresh()
            self._hyper = hyper
        self.graph.replay()
        K.note_training_step()         # parameters changed behind Tensor._version: derived images (packed weights) are stale
        self._p2p_health()
        return self._static_loss

    def _p2p_health(self):
        """Every comm.CHECK_EVERY replayed steps: did every peer-to-peer SyncBatchNorm exchange find its peers?  The check is
        a collective of its own (device sync + MAX all-reduce of the error words): every rank raises together when any of them saw a
        time-out or an overwritten slot -- a dead channel returns NaN statistics, the run must stop rather than train on them."""
        from . import comm
        if comm.CHECK_EVERY > 0 and comm.p2p_active() and self.calls % comm.CHECK_EVERY == 0:
            comm.p2p_check(what=f"training steps {self.calls - comm.CHECK_EVERY + 1}..{self.calls}")

    def _check_p2p_transport(self):
        """Before the step is frozen into a graph: did every peer answer every peer-to-peer SyncBatchNorm exchange of the eager warm-up
        steps (csrc/p2p.hip reports a peer that never showed up through an error word, it does not hang)?  The ranks agree (MIN); if
        any of them saw a time-out ALL of them close the mailboxes and the capture -- and every later step -- uses the collectives."""
        from . import comm
        if not comm.p2p_active():
            return
        ok = comm.p2p_ok()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item()) == 1
        if not ok:
            sys.stderr.write("[npp_amd.TrainStep] a peer-to-peer SyncBatchNorm exchange timed out during warm-up: every rank switches to "
                             "the collective transport (the statistics of the warm-up steps were local to each rank)\n")
            comm.disable_p2p()
            comm._p2p_tried.update({id(None), id(dist.group.WORLD)} if dist.is_initialized() else {id(None)})
            K.P2P_DIRECT = False

    def _abandon_capture(self, failed_here: bool):
        """Back to the eager path for good (on every rank alike, see __call__)."""
        self.graph, self.use_graph, self._static_in = None, False, None
        # the half-captured autograd graph pins AccumulateGrad nodes bound to the dead capture streams: let it go
        self._static_loss = None
        self.optimizer.zero_grad(set_to_none=True)
        if self.reducer is not None:
            self.reducer.reset()       # bucket counters of the abandoned backward
        import gc
        gc.collect()
        K.GRAPH_TOPOLOGY = False
        if failed_here:
            K.forget_streams()         # side streams that had joined the capture stay in capture mode: use fresh ones
        K.reset_pools()
        K.drop_pending()
        try:
            torch.cuda.synchronize()
        except Exception:      # noqa: BLE001 -- the failed capture's error may surface once more here
            pass
        from ._lib import lib as _lib_handle
        _lib_handle().npp_clear_hip_error()     # HIP's sticky last error would fail the next launch check

    def _all_ranks_ok(self, ok_here: bool) -> bool:
        """True iff EVERY rank of the default process group reports ok_here (single process: ok_here)."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() <= 1:
            return ok_here
        if dist.get_backend() == "nccl":
            flag = torch.tensor([1 if ok_here else 0], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
        else:
            flag = torch.tensor([1 if ok_here else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    @property
    def static_inputs(self):
        """The graph's input buffers (None before capture): a loader may fill them in place to skip the per-call copy."""
        return self._static_in

    @property
    def graphed(self) -> bool:
        return self.graph is not None


class _FrozenPass(TrainStep):
    """A TrainStep whose forward runs with a set of parameters frozen (requires_grad False): their gradients are neither
    computed (no weight-gradient / BatchNorm-parameter kernels are launched for them) nor accumulated."""

    def __init__(self, *a, frozen=(), **kw):
        super().__init__(*a, **kw)
        self._frozen = [p for p in frozen]

    def _thaw(self, flags):
        for p, f in zip(self._frozen, flags):
            p.requires_grad_(f)

    def _eager(self, images, labels_par, labels_pose, pose_weight):
        flags = [p.requires_grad for p in self._frozen]
        for p in self._frozen:
            p.requires_grad_(False)
        try:
            return super()._eager(images, labels_par, labels_pose, pose_weight)
        finally:
            self._thaw(flags)


class _AlphaPass(_FrozenPass):
    """The architecture pass of train_with_alpha (core/function.py:546-616): loss2 = 2 * mean(losses_par + losses_pose
    [+ 2 * model.loss_entropy() after epoch 70]) on the mini-loader batch, stepped by the architecture optimizer."""

    def __init__(self, *a, entropy=False, **kw):
        super().__init__(*a, **kw)
        self.entropy = entropy

    def _loss(self, images, labels_par, labels_pose, pose_weight):
        output_pose, output_par = self.model(images)
        losses2 = self.criterion_par(output_par, labels_par).unsqueeze(0)
        if pose_weight is not None:
            losses2 = losses2 + self.criterion_pose(output_pose, labels_pose, target_weight=pose_weight).unsqueeze(0)
        else:
            losses2 = losses2 + self.criterion_pose(output_pose, labels_pose).unsqueeze(0)
        if self.entropy:
            losses2 = losses2 + 2 * self.model.loss_entropy()
        return 2 * losses2.mean()


class SearchStep:
    """One iteration of the bi-level search loop `train_with_alpha` (core/function.py:485-621) as a callable:

        weights pass : model(images1) -> criterion_par + criterion_pose -> mean -> optimizer (all weights)          :499-531
        alpha pass   : model(images2) -> 2 * mean(losses [+ 2 * loss_entropy()]) -> a_optimizer (alphas, betas)     :546-616

    Both passes are TrainSteps (eager warm-up, then one hipGraph each).  What differs from the reference's arithmetic: nothing;
    what differs in work: the weights pass does not compute the architecture gradients the reference throws away
    (`a_optimizer.zero_grad()` before the alpha backward), and the alpha pass does not compute the ~5000 weight gradients it
    never uses (`optimizer.zero_grad()` before the next weights backward) -- each pass freezes the other pass's parameters.
    `reducer` / `a_reducer`: GradReducers over the weights / the architecture tensors (DDP averages both, search_lip_sync.py)."""

    def __init__(self, model, criterion_pose, criterion_par, optimizer, a_optimizer, reducer=None, a_reducer=None,
                 graph: Optional[bool] = None, warmup: int = 2):
        arch = list(model.arch_parameters())
        arch_ids = {id(a) for a in arch}
        weights = [p for p in list(model.parameters()) + list(criterion_pose.parameters()) + list(criterion_par.parameters())
                   if id(p) not in arch_ids and p.requires_grad]
        self.weights_pass = _FrozenPass(model, criterion_pose, criterion_par, optimizer, reducer=reducer, graph=graph,
                                        warmup=warmup, frozen=arch)
        self._alpha_args = (model, criterion_pose, criterion_par, a_optimizer)
        self._alpha_kw = dict(reducer=a_reducer, graph=graph, warmup=warmup, frozen=weights)
        self._alpha = {}

    def alpha_pass(self, entropy: bool) -> _AlphaPass:
        st = self._alpha.get(bool(entropy))
        if st is None:
            st = self._alpha[bool(entropy)] = _AlphaPass(*self._alpha_args, entropy=bool(entropy), **self._alpha_kw)
        return st

    def __call__(self, batch1, batch2, entropy: bool = False):
        """batch = (images, labels_par, labels_pose[, pose_weight]); entropy = `epoch > 70` of core/function.py:608."""
        loss1 = self.weights_pass(*batch1)
        loss2 = self.alpha_pass(entropy)(*batch2)
        return loss1, loss2
