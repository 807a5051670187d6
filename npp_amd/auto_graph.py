"""Transparent hipGraph replay of `Network.forward` + its backward for a launcher that is NOT changed.

`augment_lip_sync.py` / `core/function.py:72-107` call `model(images)`, the criteria, `loss.backward()` and
`optimizer.step()` themselves, so `train_step.TrainStep` (which owns the whole loop body) cannot be used without editing them;
issued eagerly the ~5000 launches of a step are host-bound (131 img/s).  With `NPP_AUTO_GRAPH=1` (or
`npp_amd.install_as_reference_modules(auto_graph=True)`) a training `Network` watches its calls: the first ones run eagerly
(pools, packed weights, allocator warm), then -- for the input signature seen -- the forward is captured into one hipGraph and
the backward of all outputs w.r.t. all parameters into a second one with a pool of its own (the scheme of
torch.cuda.make_graphed_callables, with this package's stream topology, scratch pools and capture-failure recovery), and
`forward` becomes an autograd node that copies the batch in, replays, and hands out the static outputs; its backward copies
the output gradients in, replays, and hands the static parameter gradients to autograd (DDP / GradReducer hooks fire as usual).
The criteria and the optimizer stay eager (a few hundred launches).

Limits, all checked at run time with a fallback to the eager forward: training mode with grad enabled only; a batch of
another shape / dtype runs eagerly; the set of parameters that require grad must not change; the outputs of one call must be
consumed before the next call (they are static buffers); a second backward before zero_grad(set_to_none=True) gets cloned
gradients (slower, still correct)."""
from __future__ import annotations

import gc
import os
import sys
import weakref

import torch
import torch.distributed as dist
from torch.autograd import Function
from torch.nn.utils import stateless as _stateless

from . import _ops as K

ENABLED = os.environ.get("NPP_AUTO_GRAPH", "0") == "1"
WARMUP_CALLS = int(os.environ.get("NPP_AUTO_GRAPH_WARMUP", "2"))


def _flatten(pose_list, par_list):
    return [t for pair in pose_list for t in pair] + [t for pair in par_list for t in pair]


def _unflatten(flat, n_pose):
    it = iter(flat)
    pose = [[next(it), next(it)] for _ in range(n_pose)]
    par = [[a, next(it)] for a in it]
    return pose, par


def _privatize_grads(g) -> bool:
    """Give every parameter whose .grad still IS a static gradient buffer of the graph a private copy (AccumulateGrad adopts
    the tensor handed out by _Replay.backward); True if any parameter holds a gradient at all."""
    held = False
    with torch.no_grad():
        for p, b in zip(g.params, g.static_grads):
            gr = p.grad
            if gr is not None:
                held = True
                if b is not None and gr.data_ptr() == b.data_ptr():
                    p.grad = gr.clone()
    return held


class _Replay(Function):
    @staticmethod
    def forward(ctx, holder, x, *params):
        g = holder.graph
        g.static_x.copy_(x)
        g.fwd.replay()
        ctx.g = g
        K.note_training_step()
        return tuple(o.detach() for o in g.static_outs)

    @staticmethod
    def backward(ctx, *grads):
        g = ctx.g
        with torch.no_grad():
            for s, gr in zip(g.static_grad_outs, grads):
                if gr is None:
                    s.zero_()
                else:
                    s.copy_(gr)
        # a second backward before zero_grad(set_to_none=True): decided BEFORE the replay, and a p.grad that still IS the graph's
        # static gradient buffer (AccumulateGrad adopted it after the previous backward) gets a private copy first -- the replay
        # overwrites that buffer, and with it the accumulated gradient (2*g2 instead of g1 + g2)
        accumulating = _privatize_grads(g)
        g.bwd.replay()
        outs = []
        for b, z in zip(g.static_grads, g.zeros_for_unused):
            if b is None:
                outs.append(z)                   # None, or a zero tensor where DDP waits for every parameter's hook
            else:
                outs.append(b.clone() if accumulating else b.detach())
        return (None, None) + tuple(outs)


class _Graph:
    pass


class AutoGraph:
    """Per-Network state: call counting, capture, replay, fallback."""

    def __init__(self, net):
        self._net = weakref.ref(net)       # (no reference cycle: captured graphs must die with the network, by refcount -- a cyclic
        self.graph = None                  #  collection that destroys a hipGraph or its pool DURING someone's capture aborts)
        self.sig = None
        self.calls = 0
        self.dead = False        # capture failed once: stay eager

    @property
    def net(self):
        return self._net()

    def _signature(self, x):
        return (tuple(x.shape), x.dtype, x.device, tuple(p.requires_grad for p in self.net.parameters()))

    def __call__(self, x):
        net = self.net
        if (self.dead or not net.training or not torch.is_grad_enabled() or not x.is_cuda
                or torch.cuda.is_current_stream_capturing() or getattr(net, "_auto_graph_off", False)):
            return net._forward_eager(x)
        sig = self._signature(x)
        if self.graph is not None and sig == self.sig:
            g = self.graph
            flat = _Replay.apply(self, x, *g.params)
            return _unflatten(list(flat), g.n_pose)
        if self.graph is not None:
            return net._forward_eager(x)          # another batch shape: eager (the captured one keeps its graph)
        if sig != self.sig:
            self.sig, self.calls = sig, 0
        self.calls += 1
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

        def agree(flag: bool) -> bool:      # MIN over the ranks: every rank takes the same path (the collective order depends on it)
            if not multi:
                return flag
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=x.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))

        # The stream topology that gets captured is scoped to this network's warm-up / capture calls (other eager work -- eval
        # forwards, other networks, a GradReducer's side stream -- keeps the eager topology).  The capture decision is collective:
        # a rank that restarted its warm-up (another batch shape) keeps every rank eager for this call.
        ready = agree(self.calls > WARMUP_CALLS)
        prev_topology = K.GRAPH_TOPOLOGY
        K.GRAPH_TOPOLOGY = True
        try:
            if not ready:
                return net._forward_eager(x)      # (its backward finds K._hub_stream as this forward left it)
            ok = agree(self._capture(x))
            K._hub_stream = None                  # the capture's origin stream: nothing eager may be sent there later
        finally:
            K.GRAPH_TOPOLOGY = prev_topology
        if not ok:
            self.graph, self.dead = None, True
            return net._forward_eager(x)
        g = self.graph
        flat = _Replay.apply(self, x, *g.params)
        return _unflatten(list(flat), g.n_pose)

    def _capture(self, x) -> bool:
        net = self.net
        g = _Graph()
        g.params = [p for p in net.parameters() if p.requires_grad]
        g.static_x = x.detach().clone()
        torch.cuda.synchronize()
        K.reset_pools()
        origin = torch.cuda.current_stream()
        # Two pools: the parameter gradients the backward graph hands out are adopted by AccumulateGrad as p.grad and may be held
        # across the NEXT forward replay (an accumulation window; a launcher that calls zero_grad after the forward,
        # core/function.py:105) -- with one shared pool that replay reuses their memory for its temporaries
        pool, pool_bwd = torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle()
        g.fwd, g.bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()          # no cyclic collection inside the capture: freeing a graph / pinned block there is an illegal call
        try:
            try:
                # Gradients are taken w.r.t. ALIASES of the parameters made inside the capture, not the parameters themselves: a
                # leaf's AccumulateGrad node is bound to the stream it was created on -- the default stream of the eager warm-up
                # steps, kept alive by whatever the launcher still holds of its last iteration (`loss`) -- and the engine
                # would synchronise that (legacy, uncapturable) stream with the capture (hipStreamEndCapture then crashes).
                named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
                with torch.cuda.graph(g.fwd, pool=pool, capture_error_mode="thread_local"):
                    aliases = {n: p.view_as(p) for n, p in named}
                    with _stateless._reparametrize_module(net, aliases):
                        pose_list, par_list = net._forward_eager(g.static_x)
                g.n_pose = len(pose_list)
                outs = _flatten(pose_list, par_list)
                g.static_grad_outs = [torch.zeros_like(o) for o in outs]
                alias_list = [aliases[n] for n, _ in named]
                # fresh scratch chunks for the backward graph: a 1x1 conv's gradient IS a slice of a pre-zeroed chunk, and the fill
                # that zeroes a chunk belongs to the graph that allocated it -- a chunk of the forward graph would be zeroed by the
                # next forward replay while p.grad still points into it
                K.reset_pools()
                with torch.cuda.graph(g.bwd, pool=pool_bwd, capture_error_mode="thread_local"):
                    grads = torch.autograd.grad(outs, alias_list, grad_outputs=g.static_grad_outs, allow_unused=True)
                    K.join_capturing_side_streams()
            except BaseException:
                torch.cuda.set_stream(origin)      # torch.cuda.graph.__exit__ raises before restoring it (train_step.py)
                raise
            finally:
                K.reset_pools()                    # chunks handed out during capture belong to the graphs' pool
                if gc_was:
                    gc.enable()
        except Exception as e:      # noqa: BLE001
            sys.stderr.write(f"[npp_amd.auto_graph] capture failed ({type(e).__name__}: {str(e)[:200]}); staying eager\n")
            del g
            gc.collect()
            K.forget_streams()
            K.reset_pools()
            try:
                torch.cuda.synchronize()
            except Exception:      # noqa: BLE001
                pass
            from ._lib import lib
            lib().npp_clear_hip_error()
            return False
        g.static_outs = outs
        g.static_grads = list(grads)
        want_zero = dist.is_available() and dist.is_initialized() and os.environ.get("NPP_AUTO_GRAPH_ZERO_UNUSED", "1") == "1"
        g.zeros_for_unused = [torch.zeros_like(p) if (b is None and want_zero) else None for p, b in zip(g.params, g.static_grads)]
        self.graph = g
        return True
