"""Fixed-genotype NPPNet on the HIP kernels -- drop-in for the reference's `models/model_augment.py`.

Same constructor (`Network(cfg, steps=4, multiplier=4, stem_multiplier=4)`, reading
`cfg.DATASET.NUM_CLASSES/NUM_JOINTS`, `cfg.TRAIN.LAYERS/INIT_CHANNELS`,
`cfg.MODEL.DECONV_WITH_BIAS/HEAD/REFINE_LAYERS`, model_augment.py:236-242), same module tree and
therefore the same `state_dict()` keys and `named_parameters()` prefixes (`cells1.`, `cells2`, `stem`,
augment_lip_sync.py:193-202), same `forward(x) -> (pose_list, par_list)` nesting, `_init_params()` and
`load_pretrain_backbone(path)`.

What differs is everything underneath: activations live in NHWC (logical NCHW, channels-last strides)
in the compute dtype (f32 = parity mode, bf16 = throughput mode, `set_compute_dtype`), every cell step
is two fused conv launches plus ONE BN-apply+add launch, and no ATen convolution / batch-norm is used.
"""
from __future__ import annotations

import contextlib
import os

import torch
import torch.nn as nn

from . import _ops as K
from . import genotypes as gt
from ._ops import BnSide
from .operations import OPS, FactorizedReduce, ReLUConvBN, fused_sum, fused_sum_apply, fused_sum_pending, fused_sum_stages, _use_batch_stats, group_wide_edges, WideEdges, group_se_pairs  # noqa: F401
from .operations import *  # noqa: F401,F403  (reference does `from models.operations import *`)

BN_MOMENTUM = 0.1
_RESAMPLE_SWAP = os.environ.get("NPP_RESAMPLE_SWAP", "1") != "0"

_compute_dtype = torch.float32


def set_compute_dtype(dtype):
    """torch.float32 (exact-f32 MFMA, parity mode) or torch.bfloat16 (bf16 storage, f32 accumulate)."""
    global _compute_dtype
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    _compute_dtype = dtype


def get_compute_dtype():
    return _compute_dtype


class Interpolate(nn.Module):
    """model_augment.py:109-116: bilinear, align_corners=True, by scale factor."""

    def __init__(self, scale_factor, mode='bilinear'):
        super().__init__()
        self.s = scale_factor
        self.mode = mode

    def forward(self, x):
        return K.interpolate_scale(x, self.s)


# ---- sequential containers that keep the reference's child indices but run fused ------------------
class _Stem(nn.Sequential):
    """Conv3x3 - BN [- ReLU]  (model_augment.py:244-272)."""

    def forward(self, x):
        conv, bn = self[0], self[1]
        y, st = K.conv2d(x, conv.weight, None, conv.stride, conv.padding, 1, relu_in=False,
                         want_stats=_use_batch_stats(bn))
        return K.bn_add(BnSide(y, bn, st), None, relu=len(self) > 2, training=bn.training)


class _Layer(nn.Sequential):
    """ReLU - Conv1x1(bias) - BN  (model_augment.py:332-351)."""

    _wide = None      # (WideEdges, index): the other layer that reads the same concatenated features (Network.__init__)

    def _cb(self):
        return self[1], self[2]

    def forward(self, x):
        conv, bn = self[1], self[2]
        side = None
        if self._wide is not None and K.WIDE:
            side = self._wide[0].pending(self._wide[1], x)      # both layers' data gradients as ONE conv (K = 512 + 384)
        if side is None:
            lvl = _use_batch_stats(bn)
            y, st = K.conv2d(x, conv.weight, conv.bias, 1, conv.padding, 1, relu_in=True, want_stats=lvl, bias_dead=lvl > 0)
            side = BnSide(y, bn, st)
        return K.bn_add(side, None, relu=False, training=bn.training)


class _Head(nn.Sequential):
    """ReLU - Conv(k, bias?) - BN - ReLU - Conv1x1(bias)  (model_augment.py:365-398)."""

    _wide = None      # (WideEdges, index): the first conv shares its input with the refinement cells' preprocess convs (Network.__init__)

    def _cb(self):
        return self[1], self[2]

    def forward(self, x):
        c1, bn, c2 = self[1], self[2], self[4]
        side = None
        if self._wide is not None and K.WIDE:
            side = self._wide[0].pending(self._wide[1], x)
        if side is None:
            lvl = _use_batch_stats(bn)
            y, st = K.conv2d(x, c1.weight, c1.bias, 1, c1.padding, 1, relu_in=True, want_stats=lvl, bias_dead=lvl > 0)
            side = BnSide(y, bn, st)
        y = K.bn_add(side, None, relu=True, training=bn.training)
        out, _ = K.conv2d(y, c2.weight, c2.bias, 1, 0, 1, relu_in=False)
        return out


class _ResampleConv(nn.Sequential):
    """Interpolate(scale) - Conv1x1(bias): the `extra_conv` of the cross-task edges (model_augment.py:590-595)."""

    def forward(self, x):
        if self[0].s > 1 and _RESAMPLE_SWAP:
            # up-sampling edge (the decoder-stage cross-task edges, model_augment.py:626-649: up to x8 from 1024 channels at 12 x 12 to
            # 128 at 96 x 96): the 1x1 conv mixes channels, the bilinear resampling mixes pixels with weights that sum to 1 -- they
            # commute, bias included -- so the conv runs on the SMALL map and the (fewer) output channels are resampled: the x8 edge
            # was a 302 MB interpolation result feeding a K = 1024 conv at 96 x 96 (and their backward passes and weight gradient)
            y, _ = K.conv2d(x, self[1].weight, self[1].bias, 1, 0, 1, relu_in=False)
            return self[0](y)
        y = self[0](x)
        out, _ = K.conv2d(y, self[1].weight, self[1].bias, 1, 0, 1, relu_in=False)
        return out


# ---- cells -------------------------------------------------------------------------------------------
class _DagCell(nn.Module):
    """Shared body of Cell / Upsample / PoseCell1 / ParCell1: `steps` nodes, each the sum of two ops
    applied to earlier states (model_augment.py:48-62, 92-106, 153-172, 210-229)."""

    def _build(self, C, edges, concat, stride_of, wrap_of):
        names, indices = zip(*edges)
        assert len(names) == len(indices)
        self._steps = len(names) // 2
        self._concat = concat
        self.multiplier = len(concat)
        self._ops = nn.ModuleList()
        for name, index in zip(names, indices):
            op = OPS[name](C, stride_of(index), True)
            scale = wrap_of(index)
            if scale is not None:
                op = nn.Sequential(op, Interpolate(scale_factor=scale))
            self._ops += [op]
        self._indices = indices
        # edges that apply the same ReLU-conv-BN to the same state run as ONE conv (operations.WideEdges)
        self._wide_groups = group_wide_edges(self._ops, names, indices, [stride_of(i) for i in indices])
        self._se_pairs = group_se_pairs(self._ops, names, indices)      # two `se_connect` on one state: one launch pair for both

    def _run(self, states, concat=None):
        """Runs the nodes; with `concat` (state indices, all of them node outputs) also returns their concatenation, whose
        parts the nodes write in place (K.ConcatBuffer)."""
        first = len(states)
        cb = None
        if concat is not None and all(i >= first for i in concat) and len(set(concat)) == len(concat):
            cb = K.ConcatBuffer(len(concat))
            pos = {idx: k for k, idx in enumerate(concat)}
        if (K.SYNC_WAVES or K.BN_MULTI) and K.helper_stream_for_edge() is None:
            # Every node whose inputs exist runs its two edges BEFORE any of their BatchNorms is applied (nodes 2 + 3, then 4 + 5 of
            # an encoder cell, model_augment.py:48-62).  SyncBatchNorm: the statistics of the whole wave travel in ONE exchange.
            # Local BatchNorm: the applies of the wave are ONE launch (K.bn_add_multi -> npp_affine_add_fin_multi) and so are their
            # backward reduces and applies -- the wave's nodes are independent and equal in shape.
            local = not K.SYNC_WAVES
            i = 0
            while i < self._steps:
                have, j, wave = len(states), i, []
                while j < self._steps and self._indices[2 * j] < have and self._indices[2 * j + 1] < have:
                    wave.append(fused_sum_pending(self._ops[2 * j], states[self._indices[2 * j]],
                                                  self._ops[2 * j + 1], states[self._indices[2 * j + 1]]))
                    j += 1
                k = 0
                while k < len(wave):
                    slot = lambda n_: cb.slot(pos[n_]) if cb is not None and n_ in pos else None      # noqa: E731
                    if local and k + 1 < len(wave):           # the rest of the wave (at most 4 nodes) as one autograd node, one launch
                        grp = wave[k:k + K.BN_MULTI_MAX]
                        n0 = len(states)
                        states.extend(K.bn_add_multi([(a_, b_, False, a_.bn.training if a_.bn is not None else False, slot(n0 + q))
                                                      for q, (a_, b_) in enumerate(grp)]))
                        k += len(grp)
                    elif K.BN_PAIRS and k + 1 < len(wave):    # two nodes of the wave as one autograd node (one backward exchange)
                        (a0, b0), (a1, b1) = wave[k], wave[k + 1]
                        t0 = a0.bn.training if a0.bn is not None else False
                        t1 = a1.bn.training if a1.bn is not None else False
                        n0 = len(states)
                        states.extend(K.bn_add_pair((a0, b0, False, t0, slot(n0)), (a1, b1, False, t1, slot(n0 + 1))))
                        k += 2
                    else:
                        a, b = wave[k]
                        states.append(fused_sum_apply(a, b, out=slot(len(states))))
                        k += 1
                i = j
        else:
            for i in range(self._steps):
                i1, i2 = self._indices[2 * i], self._indices[2 * i + 1]
                out = cb.slot(pos[len(states)]) if cb is not None and len(states) in pos else None
                states.append(fused_sum(self._ops[2 * i], states[i1], self._ops[2 * i + 1], states[i2], out=out))
        if concat is None:
            return states
        parts = [states[i] for i in concat]
        return states, (cb.result(parts) if cb is not None else K.concat(parts))


def _preprocess_pair(pre0, s0, pre1, s1):
    """[pre0(s0), pre1(s1)] with both convs issued before either BatchNorm is applied: under SyncBatchNorm the statistics of the two
    then travel in ONE exchange (they sit back to back in the stream's statistics pool and the first apply flushes both)."""
    from .operations import pending_of
    a, b = pending_of(pre0, s0), pending_of(pre1, s1)
    if ((K.SYNC_WAVES and K.BN_PAIRS) or (not K.SYNC_WAVES and K.BN_MULTI)) and a.bn is not None and b.bn is not None:
        # ... and as ONE autograd node, so that their backward passes share one exchange as well (_ops._BnAddPair); with local
        # statistics the two applies -- and the two backward passes -- are one launch each
        return list(K.bn_add_pair((a, None, False, a.bn.training, None), (b, None, False, b.bn.training, None)))
    outs = []
    for side in (a, b):
        outs.append(K.bn_add(side, None, relu=False, training=side.bn.training) if side.bn is not None else side.x)
    return outs


class Cell(_DagCell):
    """Encoder cell, model_augment.py:16-62."""

    def __init__(self, genotype, C_prev_prev, C_prev, C, reduction, reduction_prev):
        super().__init__()
        if reduction_prev:
            self.preprocess0 = FactorizedReduce(C_prev_prev, C)
        else:
            self.preprocess0 = ReLUConvBN(C_prev_prev, C, 1, 1, 0, affine=True)
        self.preprocess1 = ReLUConvBN(C_prev, C, 1, 1, 0, affine=True)
        edges, concat = (genotype.reduce, genotype.reduce_concat) if reduction else \
            (genotype.normal, genotype.normal_concat)
        self._build(C, edges, concat, lambda idx: 2 if reduction and idx < 2 else 1, lambda idx: None)

    def forward(self, s0, s1):
        _, out = self._run(_preprocess_pair(self.preprocess0, s0, self.preprocess1, s1), self._concat)
        return out

    def stages(self, s0, s1, result):
        """forward() as a generator that pauses after every stage (each preprocess, each node): Network.forward steps the
        two branches' cells alternately, so that the host issues -- and, under SyncBatchNorm, the one communicator orders --
        their work interleaved at node granularity instead of cell by cell.  result: 1-element list for the output."""
        states = [self.preprocess0(s0)]
        yield
        states.append(self.preprocess1(s1))
        yield
        concat = self._concat
        first = len(states)
        cb = pos = None
        if all(i >= first for i in concat) and len(set(concat)) == len(concat):
            cb = K.ConcatBuffer(len(concat))
            pos = {idx: k for k, idx in enumerate(concat)}
        for i in range(self._steps):
            i1, i2 = self._indices[2 * i], self._indices[2 * i + 1]
            out = cb.slot(pos[len(states)]) if cb is not None and len(states) in pos else None
            node = [None]
            yield from fused_sum_stages(self._ops[2 * i], states[i1], self._ops[2 * i + 1], states[i2], node, out=out)
            states.append(node[0])
            if i + 1 < self._steps:
                yield
        parts = [states[i] for i in concat]
        result[0] = cb.result(parts) if cb is not None else K.concat(parts)


class Upsample(_DagCell):
    """Decoder cell, model_augment.py:64-106: ops fed from input 0 are followed by a x2 bilinear."""

    def __init__(self, upsample, upsample_concat, C_prev_prev, C_prev):
        super().__init__()
        self.preprocess0 = ReLUConvBN(C_prev_prev, C_prev // 4, 1, 1, 0, affine=True)
        self.preprocess1 = ReLUConvBN(C_prev, C_prev // 4, 1, 1, 0, affine=True)
        self._build(C_prev // 4, upsample, upsample_concat, lambda idx: 1, lambda idx: 2 if idx == 0 else None)

    def forward(self, s0, s1):
        _, out = self._run(_preprocess_pair(self.preprocess0, s0, self.preprocess1, s1), self._concat)
        return out


class _FuseCell(_DagCell):
    def __init__(self, edges, concat, C_prev_prev, C_prev, C_cur, order):
        super().__init__()
        self.order = order
        if order == 0:
            self.preprocess0 = ReLUConvBN(C_prev_prev, C_cur, 1, 1, 0, affine=True)
            self.preprocess1 = ReLUConvBN(C_prev, C_cur, 1, 1, 0, affine=True)
            self.preprocess2 = ReLUConvBN(C_cur, C_cur, 1, 1, 0, affine=True)
        else:
            self.preprocess0 = ReLUConvBN(3 * C_cur, C_cur, 1, 1, 0, affine=True)
            self.preprocess1 = ReLUConvBN(4 * C_cur, C_cur, 1, 1, 0, affine=True)
            self.preprocess2 = ReLUConvBN(4 * C_cur, C_cur, 1, 1, 0, affine=True)
        wrap = (lambda idx: {0: 4, 1: 2}.get(idx)) if order == 0 else (lambda idx: None)
        self._build(C_cur, edges, concat, lambda idx: 1, wrap)

    def forward(self, s0, s1, s2, foreign=None, hub=None):
        """`foreign`: index of the input produced by the other task branch; with the hub topology (Network.forward) the
        op that consumes it runs on the hub stream, so that the two branch streams never wait on each other directly."""
        if self.order == 0:
            return self._forward_order0(s0, s1, s2)
        pre = [self.preprocess0, self.preprocess1, self.preprocess2]
        ins = [s0, s1, s2]
        cb1 = None
        if hub is not None and foreign is not None:
            outs = [None, None, None]
            cur = torch.cuda.current_stream()
            K.note_stream_use(ins[foreign], hub)      # read on the hub stream, forward and backward (see operations.fused_sum)
            with torch.cuda.stream(hub):
                outs[foreign] = pre[foreign](ins[foreign])
            for i in range(3):
                if i != foreign:
                    outs[i] = pre[i](ins[i])
            cur.wait_stream(hub)
            outs[foreign].record_stream(cur)
        else:
            # the three preprocessed inputs are returned concatenated (fea1): they are written into that buffer directly
            cb1 = K.ConcatBuffer(3) if all(isinstance(m, ReLUConvBN) for m in pre) else None
            if cb1 is not None and K.BN_MULTI and not K.SYNC_WAVES:
                # the three convs first, then their BatchNorm applies as ONE launch (and one reduce + one apply launch backward)
                from .operations import pending_of
                sides = [pending_of(pre[i], ins[i]) for i in range(3)]
                outs = K.bn_add_multi([(sides[i], None, False, sides[i].bn.training, cb1.slot(i)) for i in range(3)])
            else:
                outs = [pre[i](ins[i], out=cb1.slot(i)) if cb1 is not None else pre[i](ins[i]) for i in range(3)]
        st, fea2 = self._run(outs, self._concat)
        fea1 = cb1.result(st[0:3]) if cb1 is not None else K.concat(st[0:3])
        return fea1, fea2

    def _forward_order0(self, s0, s1, s2):
        """order == 0 (model_augment.py:119-172, 174-229; never built by the reference's Network, model_augment.py:357-363): the inputs
        arrive at 1/4, 1/2 and full resolution, the ops on inputs 0 / 1 are followed by a bilinear x4 / x2 (`wrap` in __init__), and after
        the node loop states 0 and 1 are replaced by F.interpolate(scale_factor=4 / 2) in its default NEAREST mode before both
        concatenations (model_augment.py:167-171)."""
        states = self._run([self.preprocess0(s0), self.preprocess1(s1), self.preprocess2(s2)], None)
        states = list(states)
        states[0] = K.nearest(states[0], 4)
        states[1] = K.nearest(states[1], 2)
        return K.concat(states[0:3]), K.concat([states[i] for i in self._concat])

    def stages(self, s0, s1, s2, foreign, hub, result):
        """forward() as a generator pausing after every preprocess and every edge (see Cell.stages); result[0] = (fea1, fea2)."""
        if self.order == 0:      # (no lockstep form: one stage)
            result[0] = self._forward_order0(s0, s1, s2)
            return
        pre = [self.preprocess0, self.preprocess1, self.preprocess2]
        ins = [s0, s1, s2]
        cb1 = K.ConcatBuffer(3) if all(isinstance(m, ReLUConvBN) for m in pre) else None
        outs = [None, None, None]
        order = [i for i in range(3) if i != foreign] + ([foreign] if foreign is not None else [])
        for i in order:
            slot = cb1.slot(i) if cb1 is not None else None
            if hub is not None and i == foreign:
                cur = torch.cuda.current_stream()
                K.note_stream_use(ins[i], hub)        # read on the hub stream, forward and backward (see operations.fused_sum)
                with torch.cuda.stream(hub):
                    outs[i] = pre[i](ins[i], out=slot) if slot is not None else pre[i](ins[i])
                cur.wait_stream(hub)
                outs[i].record_stream(cur)
                K.note_stream_use(outs[i], hub)       # (a ConcatBuffer slot: the buffer was allocated on `cur` and is written on the hub)
            else:
                outs[i] = pre[i](ins[i], out=slot) if slot is not None else pre[i](ins[i])
            yield
        states = list(outs)
        concat = self._concat
        first = len(states)
        cb = pos = None
        if all(i >= first for i in concat) and len(set(concat)) == len(concat):
            cb = K.ConcatBuffer(len(concat))
            pos = {idx: k for k, idx in enumerate(concat)}
        for i in range(self._steps):
            i1, i2 = self._indices[2 * i], self._indices[2 * i + 1]
            out = cb.slot(pos[len(states)]) if cb is not None and len(states) in pos else None
            node = [None]
            yield from fused_sum_stages(self._ops[2 * i], states[i1], self._ops[2 * i + 1], states[i2], node, out=out)
            states.append(node[0])
            if i + 1 < self._steps:
                yield
        parts = [states[i] for i in concat]
        fea2 = cb.result(parts) if cb is not None else K.concat(parts)
        fea1 = cb1.result(states[0:3]) if cb1 is not None else K.concat(states[0:3])
        result[0] = (fea1, fea2)


class PoseCell1(_FuseCell):
    """model_augment.py:119-172"""

    def __init__(self, pose, pose_concat, C_prev_prev, C_prev, C_cur, order):
        super().__init__(pose, pose_concat, C_prev_prev, C_prev, C_cur, order)


class ParCell1(_FuseCell):
    """model_augment.py:176-229"""

    def __init__(self, par, par_concat, C_prev_prev, C_prev, C_cur, order):
        super().__init__(par, par_concat, C_prev_prev, C_prev, C_cur, order)


# ---- network -----------------------------------------------------------------------------------------
_side_streams = {}
K._stream_caches.append(_side_streams)


def _side_stream(device, which=0):
    key = (device.type, device.index, which)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
        if which == 0:
            K._branch_b_streams.add(st.cuda_stream)
    return st


_DONE = object()


def _alternate(g1, ctx1, g2, ctx2):
    """Step two stage generators alternately, each under its own stream context, until both are exhausted."""
    live1 = live2 = True
    while live1 or live2:
        if live1:
            with ctx1():
                live1 = next(g1, _DONE) is not _DONE
        if live2:
            with ctx2():
                live2 = next(g2, _DONE) is not _DONE


def _lockstep(mode: int, sync_bn: bool) -> bool:
    """Step the two encoders' cells alternately, stage by stage (Cell.stages)?  NPP_LOCKSTEP=1 / 0 forces it; default: only
    when the branches run on their own streams AND SyncBatchNorm funnels their exchanges through one stream."""
    v = os.environ.get("NPP_LOCKSTEP")
    if v is not None:
        return v != "0"
    return mode == 3 and sync_bn


def _syncbn_stream_mode() -> int:
    v = os.environ.get("NPP_SYNCBN_STREAMS")
    if v in ("1", "2", "3"):
        return int(v)
    return 3 if K.GRAPH_TOPOLOGY else 1


def _stream_mode() -> int:
    """NPP_STREAMS: 1 = everything on the caller's stream; 2 (default) = one stream per task branch (the pose branch on the
    caller's); 3 = hub topology: both branches on side streams, the caller's stream runs what touches both and a share of
    the cells' second edges (NPP_HUB_SHARE, default 2/3) -- capturable, but measured slower than 2 (81-87 vs 77 ms); 4 = 2 plus per-branch helper streams (eager only, see _ops.helper_stream)."""
    v = os.environ.get("NPP_STREAMS", "2")
    return {"1": 1, "2": 2, "3": 3, "4": 2}.get(v, 2)


class Network(nn.Module):
    """model_augment.py:231-709."""

    def __init__(self, cfg, steps=4, multiplier=4, stem_multiplier=4):
        super().__init__()
        self._num_classes = cfg.DATASET.NUM_CLASSES
        self._num_joints = cfg.DATASET.NUM_JOINTS
        self._layers = cfg.TRAIN.LAYERS
        self.C = cfg.TRAIN.INIT_CHANNELS
        self.deconv_with_bias = cfg.MODEL.DECONV_WITH_BIAS
        self._head = cfg.MODEL.HEAD
        self.refine_layers = cfg.MODEL.REFINE_LAYERS
        C = self.C

        def stem(cin, cout, stride, relu):
            mods = [nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False),
                    nn.BatchNorm2d(cout, momentum=BN_MOMENTUM)]
            if relu:
                mods.append(nn.ReLU(inplace=True))
            return _Stem(*mods)

        self.stem0, self.stem1, self.stem2 = stem(3, C, 2, True), stem(C, 2 * C, 2, True), stem(2 * C, 2 * C, 1, False)
        self.stem3, self.stem4, self.stem5 = stem(3, C, 2, True), stem(C, 2 * C, 2, True), stem(2 * C, 2 * C, 1, False)

        L = self._layers
        self._taps = [L // 4 - 1, 2 * L // 4 - 1, 3 * L // 4 - 1, 4 * L // 4 - 1]
        reductions = [L // 4, 2 * L // 4, 3 * L // 4]
        C_pp, C_p, C_curr = 2 * C, 2 * C, int(C / 2)
        self.cells1, self.cells2 = nn.ModuleList(), nn.ModuleList()
        self.num_inchannels = []
        red_prev = False
        for i in range(L):
            if i in self._taps:
                self.num_inchannels.append(int(C_curr * multiplier))
            red = i in reductions
            if red:
                C_curr *= 2
            self.cells1 += [Cell(gt.ENCODER, C_pp, C_p, C_curr, red, red_prev)]
            self.cells2 += [Cell(gt.ENCODER, C_pp, C_p, C_curr, red, red_prev)]
            red_prev = red
            C_pp, C_p = C_p, multiplier * C_curr
        self.num_inchannels = self.num_inchannels[::-1]
        widths = self.num_inchannels[::-1]
        # The output of cell i - 2 is read by preprocess1 of cell i - 1 AND by preprocess0 of cell i (model_augment.py:413-417: s0, s1 =
        # s1, cell(s0, s1)): inside a stage both are ReLU - Conv1x1(4C -> C) - BN, so they run as ONE conv 4C -> 2C when the first of
        # them is asked (operations.WideEdges; the second picks up its slice one cell later), and their data gradients as one conv
        # 2C -> 4C -- the sum of the state's two gradients happens in the MFMA accumulators.
        self._wide_groups = []
        for cells in (self.cells1, self.cells2):
            for i in range(1, L):
                a, b = cells[i - 1].preprocess1, cells[i].preprocess0
                if type(a) is ReLUConvBN and type(b) is ReLUConvBN and WideEdges.fits([a, b]):
                    self._wide_groups.append(WideEdges([a, b]))

        # encoder-stage cross-task edges (model_augment.py:301-309, _compile :576-599)
        self._indices1, ops = self._compile(gt.INTER.task1, widths)
        self._ops1 = nn.ModuleList(ops)
        self._indices2, ops = self._compile(gt.INTER.task2, widths)
        self._ops2 = nn.ModuleList(ops)
        # decoder-stage cross-task edges (model_augment.py:311-320, _compile3 :626-649)
        resolution = [1, 1 / 2, 1 / 4, 1 / 8, 1 / 4, 1 / 2, 1]
        channels = [int(2 * C / r) for r in resolution]
        self.up_indices1, ops = self._compile3(gt.INTER.task3, resolution, channels)
        self.up_ops1 = nn.ModuleList(ops)
        self.up_indices2, ops = self._compile3(gt.INTER.task4, resolution, channels)
        self.up_ops2 = nn.ModuleList(ops)

        self.upsamples1, self.upsamples2 = nn.ModuleList(), nn.ModuleList()
        nin = self.num_inchannels
        for j in range(len(nin) - 1):
            self.upsamples1 += [Upsample(gt.DECODER.upsample1, gt.DECODER.upsample_concat1, nin[j], nin[j + 1])]
        for j in range(len(nin) - 1):
            self.upsamples2 += [Upsample(gt.DECODER.upsample2, gt.DECODER.upsample_concat2, nin[j], nin[j + 1])]

        Cf = nin[3]

        def layer(cout):
            return _Layer(nn.ReLU(), nn.Conv2d(8 * Cf, cout, kernel_size=1, padding=0, dilation=1),
                          nn.BatchNorm2d(cout, momentum=BN_MOMENTUM))

        self.pose_layer = layer(4 * Cf)
        self.pose_auxlayer = layer(3 * Cf)
        self.par_layer = layer(4 * Cf)
        self.edge_layer = layer(3 * Cf)

        # pose_auxlayer + pose_layer read the same concatenation x1 (edge_layer + par_layer: x2), model_augment.py:540-548: the forward
        # convs keep their own launches (tiles chosen per width), their data gradients run as ONE conv over the concatenated dy --
        # the second one was a 300 MB read-add-store into the first one's result
        self._wide_groups.append(WideEdges([self.pose_auxlayer, self.pose_layer], separate_fwd=True))
        self._wide_groups.append(WideEdges([self.edge_layer, self.par_layer], separate_fwd=True))

        self.pose_net, self.par_net = nn.ModuleList(), nn.ModuleList()
        for _ in range(3):
            self.pose_net.append(PoseCell1(gt.FUSION.pose, gt.FUSION.pose_concat, Cf, Cf, Cf, 1))
            self.par_net.append(ParCell1(gt.FUSION.par, gt.FUSION.par_concat, Cf, Cf, Cf, 1))

        # The pose and the parsing refinement cell of one stage read the SAME in3 / in4 through their preprocess1 / preprocess2
        # (ReLU - Conv1x1(4C -> C) - BN, model_augment.py:555-571): one conv 512 -> 256 for both, and -- where the bytes are -- ONE data
        # gradient 256 -> 512 instead of two 151 MB results summed by a read-add-store.  The cells run on the two branch streams:
        # whichever asks first runs the conv, the other stream waits for it (operations.WideEdges.pending).
        self._cross_pairs = []
        for pc, qc in zip(self.pose_net, self.par_net):
            for a, b in ((pc.preprocess1, qc.preprocess1), (pc.preprocess2, qc.preprocess2)):
                if type(a) is ReLUConvBN and type(b) is ReLUConvBN and WideEdges.fits([a, b]):
                    self._cross_pairs.append([a, b])

        def head(cin, mid, k, cout, bias1=True):
            return _Head(nn.ReLU(), nn.Conv2d(cin, mid, kernel_size=k, padding=k // 2, dilation=1, bias=bias1),
                         nn.BatchNorm2d(mid, momentum=BN_MOMENTUM), nn.ReLU(inplace=True),
                         nn.Conv2d(mid, cout, kernel_size=1, padding=0, dilation=1, bias=True))

        self.pose_head, self.pose_auxnet = nn.ModuleList(), nn.ModuleList()
        self.par_head, self.edge_head = nn.ModuleList(), nn.ModuleList()
        for _ in range(self.refine_layers + 1):
            self.pose_head.append(head(4 * Cf, 256, 1, self._num_joints))
            self.pose_auxnet.append(head(3 * Cf, 128, 3, self._num_joints))
            self.par_head.append(head(4 * Cf, 256, 1, self._num_classes))
            self.edge_head.append(head(3 * Cf, 6, 3, 2, bias1=False))
        # ... and the first stage's in3 / in4 are read by pose_head[0] / par_head[0] as well (heads(0) comes before the cells,
        # model_augment.py:549-553): their 512 -> 256 conv keeps its launch, its data gradient joins the pair's (K = 128 + 128 + 256)
        for j, pair in enumerate(self._cross_pairs):
            head0 = (self.pose_head[0], self.par_head[0])[j] if j < 2 else None
            if head0 is not None and WideEdges.fits_mixed(pair + [head0]):
                self._wide_groups.append(WideEdges(pair + [head0], cross_stream=True))
            else:
                self._wide_groups.append(WideEdges(pair, cross_stream=True))
        del self._cross_pairs
        self._packer = None
        self._auto = None          # auto_graph.AutoGraph, created by the first training forward under NPP_AUTO_GRAPH=1
        self._init_params()

    # -- construction helpers ---------------------------------------------------------------------------
    @staticmethod
    def _edge(name, c_src, c_dst, scale, needs_extra):
        op = OPS[name](c_src, 1, True)
        if needs_extra:
            op = nn.Sequential(op, _ResampleConv(Interpolate(scale), nn.Conv2d(c_src, c_dst, 1)))
        return op

    def _compile(self, geno, C_list):
        indices, ops = [], []
        for cont, stage in enumerate(geno):
            names, idx = zip(*stage)
            indices.append(idx)
            for n, ind in zip(names, idx):
                ops.append(self._edge(n, C_list[ind], C_list[cont], 1 / 2 ** (cont - ind), ind != cont))
        return indices, ops

    def _compile3(self, geno, resolutions, C_list):
        indices, ops = [], []
        for cont, stage in enumerate(geno):
            names, idx = zip(*stage)
            indices.append(idx)
            for n, ind in zip(names, idx):
                ops.append(self._edge(n, C_list[ind], C_list[4 + cont], resolutions[4 + cont] / resolutions[ind],
                                      ind != 4 + cont))
        return indices, ops

    # -- forward ----------------------------------------------------------------------------------------
    @staticmethod
    def _cross(ops, base, indices, feats):
        """sum_j ops[base + j](feats[indices[j]])"""
        z = None
        for j, ind in enumerate(indices):
            y = ops[base + j](feats[ind])
            z = y if z is None else K.add(z, y)
        return z

    def forward(self, x):
        """model_augment.py:402-574.  With NPP_AUTO_GRAPH=1 a training forward (and its backward) is replayed as hipGraphs after
        the first calls (npp_amd/auto_graph.py); otherwise -- and always in eval mode, under no_grad or inside a capture -- the
        kernels are issued one by one."""
        from . import auto_graph
        if auto_graph.ENABLED and self.training:
            if self._auto is None:
                self._auto = auto_graph.AutoGraph(self)
            return self._auto(x)
        return self._forward_eager(x)

    def _forward_eager(self, x):
        if not x.is_cuda:
            raise RuntimeError("npp_amd.Network runs on the MI355X HIP kernels only: move the input to cuda "
                               "(there is no CPU fallback)")
        dt = _compute_dtype
        K.fan_reset()
        if self._packer is None:
            from .operations import SE_Block
            skip = set()
            for m in self.modules():
                if isinstance(m, SE_Block):
                    skip.update((id(m.conv1.weight), id(m.conv2.weight)))
            groups = [g for m in self.modules() for g in getattr(m, "_wide_groups", ())] if K.WIDE else []
            self._packer = K.WeightPacker((m.weight for m in self.modules()
                                           if isinstance(m, nn.Conv2d) and m.groups == 1 and id(m.weight) not in skip), groups)
        self._packer.pack_if_stale(dt, x.device, force=self.training)
        if self.training:
            K.note_training_step()      # an optimizer step probably follows: the next eval forward must repack
        x = K.image_to_nhwc(x, dt)
        # The pose branch runs on the caller's stream ("A"), the parsing branch on a side stream ("B"); the branches
        # meet at the 4 encoder taps, the 3 decoder stages and the refinement cells.  Kernels of one branch that cannot
        # fill 256 CUs (12x12 / 24x24 maps, tile tails) overlap with the other branch's; a captured hipGraph keeps the
        # two-branch structure (tools/graph_concurrency.py) and autograd replays each node's backward on the stream of
        # its forward.
        # SyncBatchNorm across ranks: every collective must sit on one stream (the hipGraph capture's origin; RCCL's own
        # launch ordering adds two-way edges between the user stream and its internal stream, which this ROCm only
        # survives on the origin).  One communicator orders its collectives totally, in host issue order: issued cell by
        # cell, a branch's exchanges wait for the whole previous cell of the other branch and the two branches run one
        # after the other (hub topology 219 img/s = single stream 217, 1-rank exercise).  So under SyncBN the branches get
        # their own streams (hub topology, 3) AND are issued in lockstep, edge by edge (Cell.stages / _alternate): the
        # exchanges alternate A, B, A, B and each one's latency hides behind the other branch's kernels -- 253 img/s.
        # That is the topology of a step that will be replayed as a hipGraph (TrainStep asks for it); issued eagerly its
        # ~2000 stream switches and event pairs cost the host more than the overlap returns (59 vs ~100 img/s), so plain
        # eager SyncBN steps stay on one stream.  NPP_SYNCBN_STREAMS=1 / 3 forces either.
        sync_bn = self._sync_bn_active()
        # Ranks of one node exchange the statistics through peer-to-peer mailboxes (csrc/p2p.hip): an exchange is then an ordinary
        # kernel on the stream that needs it, and the branches keep the two-stream topology of the local-BatchNorm step.
        K.P2P_DIRECT = bool(sync_bn and x.is_cuda and os.environ.get("NPP_SYNCBN_STREAMS") is None
                            and os.environ.get("NPP_P2P_DIRECT", "1") != "0" and self._p2p_ready())
        mode = _stream_mode() if (not sync_bn or K.P2P_DIRECT) else _syncbn_stream_mode()
        K.SYNC_WAVES = bool(sync_bn and os.environ.get("NPP_SYNC_WAVES", "1") != "0")
        two = mode >= 2
        K._helper_uses = 0
        K._hub_offload = None
        so = sa = sb = None
        if two:
            so = torch.cuda.current_stream()
            sb = _side_stream(x.device, 0)
            # 2: the pose branch shares the caller's stream.  3 (hub): both branches get their own stream; the caller's
            # stream (the hipGraph capture's origin) runs everything that touches both branches -- the cross-task
            # edges, the refinement cells' foreign inputs -- plus a share of the cells' second edges, and is the only
            # stream either branch ever waits on (a capture on this ROCm dies on two-way waits between non-origin streams)
            sa = _side_stream(x.device, 1) if mode == 3 else so
            K._hub_stream = so       # collectives (SyncBatchNorm) of both branches, forward and backward, go here
            if mode == 3:
                K._hub_offload = (so, {sa.cuda_stream, sb.cuda_stream})
        else:
            K._hub_stream = None
        hub = so if mode == 3 else None

        def on_a():
            return torch.cuda.stream(sa) if mode == 3 else contextlib.nullcontext()

        def on_b():
            return torch.cuda.stream(sb) if two else contextlib.nullcontext()

        def meet(*tensors):
            """Each stream waits for what the others have issued so far; `tensors` cross streams from here on."""
            if not two:
                return
            if mode == 3:
                so.wait_stream(sa)
                so.wait_stream(sb)
                sa.wait_stream(so)
                sb.wait_stream(so)
            else:
                ea, eb = torch.cuda.Event(), torch.cuda.Event()
                ea.record(so)
                eb.record(sb)
                so.wait_event(eb)
                sb.wait_event(ea)
            for t in tensors:
                for st in {id(so): so, id(sa): sa, id(sb): sb}.values():
                    t.record_stream(st)

        meet(x)
        with on_a():
            s1 = self.stem2(s0 := self.stem1(self.stem0(x)))
            s1 = K.stamp_through(s1, "A stems")
        with on_b():
            s3 = self.stem5(s2 := self.stem4(self.stem3(x)))
            s3 = K.stamp_through(s3, "B stems")
        f1, f2 = [], []
        k1 = k2 = stage = 0
        lockstep = _lockstep(mode, sync_bn)
        # lockstep + hub topology: the two branches' SyncBatchNorm statistics share one pool and travel in ONE collective per
        # lockstep stage (K.SYNC_MERGE; NPP_SYNC_MERGE=0 keeps one exchange per branch)
        K.SYNC_MERGE = bool(lockstep and sync_bn and mode >= 2 and os.environ.get("NPP_SYNC_MERGE", "1") != "0")
        for i, (cell1, cell2) in enumerate(zip(self.cells1, self.cells2)):
            if lockstep:
                r1, r2 = [None], [None]
                _alternate(cell1.stages(s0, s1, r1), on_a, cell2.stages(s2, s3, r2), on_b)
                s0, s1 = s1, r1[0]
                s2, s3 = s3, r2[0]
            else:
                with on_a():
                    s0, s1 = s1, cell1(s0, s1)
                with on_b():
                    s2, s3 = s3, cell2(s2, s3)
            if i in self._taps:
                if K.STAMPS is not None:
                    with on_a():
                        s1 = K.stamp_through(s1, f"A enc{stage}")
                    with on_b():
                        s3 = K.stamp_through(s3, f"B enc{stage}")
                f1.append(s1)
                f2.append(s3)
                meet(*f1, *f2)
                ind1, ind2 = self._indices1[stage], self._indices2[stage]
                if mode == 3:
                    z1 = self._cross(self._ops1, k1, ind1, f2)      # hub: reads the other branch's features
                    z2 = self._cross(self._ops2, k2, ind2, f1)
                    sa.wait_stream(so)
                    sb.wait_stream(so)
                    z1.record_stream(sa)
                    z2.record_stream(sb)
                    with on_a():
                        s1 = K.add(s1, z1)
                    with on_b():
                        s3 = K.add(s3, z2)
                else:
                    s1 = K.add(s1, self._cross(self._ops1, k1, ind1, f2))
                    with on_b():
                        s3 = K.add(s3, self._cross(self._ops2, k2, ind2, f1))
                k1 += len(ind1)
                k2 += len(ind2)
                stage += 1
                f1[-1], f2[-1] = s1, s3
        # decoder: three structurally identical stages (model_augment.py:448-533)
        k1 = k2 = 0
        for d in range(3):
            with on_a():
                o1 = self.upsamples1[d](f1[3] if d == 0 else f1[-1], f1[2 - d])
                o1 = K.stamp_through(o1, f"A dec{d}")
            with on_b():
                o2 = self.upsamples2[d](f2[3] if d == 0 else f2[-1], f2[2 - d])
                o2 = K.stamp_through(o2, f"B dec{d}")
            f1.append(o1)
            f2.append(o2)
            meet(*f1, *f2)
            ind1, ind2 = self.up_indices1[d], self.up_indices2[d]
            if mode == 3:
                z1 = self._cross(self.up_ops1, k1, ind1, f2)
                z2 = self._cross(self.up_ops2, k2, ind2, f1)
                sa.wait_stream(so)
                sb.wait_stream(so)
                z1.record_stream(sa)
                z2.record_stream(sb)
                with on_a():
                    f1[-1] = K.add(o1, z1)
                with on_b():
                    f2[-1] = K.add(o2, z2)
            else:
                g1 = list(f1)        # both cross sums read the features as they are BEFORE this stage's adds
                f1[-1] = K.add(o1, self._cross(self.up_ops1, k1, ind1, f2))
                with on_b():
                    f2[-1] = K.add(o2, self._cross(self.up_ops2, k2, ind2, g1))
            k1 += len(ind1)
            k2 += len(ind2)
        H, W = f1[0].shape[2], f1[0].shape[3]
        with on_a():
            x1 = K.concat([f1[0], f1[6], K.bilinear(f1[5], H, W), K.bilinear(f1[4], H, W)])
            x1 = K.stamp_through(x1, "A cross+cat")
            in1, in3 = self.pose_auxlayer(x1), self.pose_layer(x1)
            in3 = K.stamp_through(in3, "A layers")
        with on_b():
            x2 = K.concat([f2[0], f2[6], K.bilinear(f2[5], H, W), K.bilinear(f2[4], H, W)])
            x2 = K.stamp_through(x2, "B cross+cat")
            in2, in4 = self.edge_layer(x2), self.par_layer(x2)
            in4 = K.stamp_through(in4, "B layers")
        pose_list, par_list = [], []

        def heads(i):      # issued alternately (see _lockstep); the streams make the order irrelevant otherwise
            with on_a():
                pose_aux = self.pose_auxnet[i](in1)
            with on_b():
                edge = self.edge_head[i](in2)
            with on_a():
                pose_map = self.pose_head[i](in3)
            with on_b():
                par_map = self.par_head[i](in4)
            pose_list.append([pose_map, pose_aux])
            par_list.append([par_map, edge])

        heads(0)
        for i in range(1, self.refine_layers + 1):
            for j in range(3):
                m = 2 * (i - 1) + j
                meet(in1, in2, in3, in4)
                if lockstep:
                    r1, r2 = [None], [None]
                    _alternate(self.pose_net[m].stages(in1, in3, in4, 2, hub, r1), on_a,
                               self.par_net[m].stages(in2, in3, in4, 1, hub, r2), on_b)
                    (n1, tmp), (in2, n4) = r1[0], r2[0]
                else:
                    with on_a():
                        n1, tmp = self.pose_net[m](in1, in3, in4, foreign=2, hub=hub)     # in4 comes from the parsing branch
                    with on_b():
                        in2, n4 = self.par_net[m](in2, in3, in4, foreign=1, hub=hub)      # in3 from the pose branch
                in1, in3, in4 = n1, tmp, n4
                if K.STAMPS is not None:
                    with on_a():
                        in3 = K.stamp_through(in3, f"A refine{j}")
                    with on_b():
                        in4 = K.stamp_through(in4, f"B refine{j}")
            heads(i)
        K.stamp("fwd end (issue order)")
        K._hub_offload = None
        if K.SYNC_MERGE:
            K._shared_sync_pool.flush()      # (nothing should be waiting: every BatchNorm has been applied)
            K.SYNC_MERGE = False
        if two:      # the caller's stream owns every output from here on
            so.wait_stream(sb)
            if mode == 3:
                so.wait_stream(sa)
            for pair in par_list + (pose_list if mode == 3 else []):
                for t in pair:
                    t.record_stream(so)
        K.fan_reset()
        return pose_list, par_list

    def __getstate__(self):      # deepcopy / pickle: derived caches (packed weights, captured graphs) stay behind
        d = self.__dict__.copy()
        d["_packer"] = None
        d["_auto"] = None
        return d

    def _p2p_ready(self) -> bool:
        from . import comm
        grp = getattr(self, "_sync_pg", False)
        if grp is False:
            grp = None
            for m in self.modules():
                if isinstance(m, nn.SyncBatchNorm):
                    grp = K._sync_group(m)[0]
                    break
            self._sync_pg = grp
        return comm.ensure_p2p(grp)

    def _sync_bn_active(self) -> bool:
        import torch.distributed as dist
        if K.SYNC_OFF or not (dist.is_available() and dist.is_initialized()):
            return False
        if dist.get_world_size() <= 1 and not K._SYNC_EVEN_ALONE:
            return False
        return any(isinstance(m, nn.SyncBatchNorm) for m in self.modules())

    # -- parameter handling -------------------------------------------------------------------------------
    def _init_params(self):
        """model_augment.py:651-671: xavier-normal conv weights, zero biases, BN gamma=1 beta=0."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight.data)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.modules.batchnorm._BatchNorm):
                if m.affine:
                    m.weight.data.fill_(1)
                    m.bias.data.zero_()

    def load_pretrain_backbone(self, path=''):
        """Tolerant key-wise loader, model_augment.py:673-709: strips a leading `module.`, keeps our tensor
        where shapes disagree, ignores unknown keys, leaves missing ones untouched."""
        if not os.path.isfile(path):
            return
        loaded = torch.load(path, map_location='cpu')
        own = self.state_dict()
        merged = {}
        for k, v in loaded.items():
            kk = k[7:] if k.startswith('module') else k
            if kk in own:
                if tuple(v.shape) != tuple(own[kk].shape):
                    print('Skip loading parameter {}, required shape{}, loaded shape{}.'.format(
                        kk, tuple(own[kk].shape), tuple(v.shape)))
                    v = own[kk]
                merged[kk] = v
        for k, v in own.items():
            merged.setdefault(k, v)
        msg = self.load_state_dict(merged, strict=False)
        print("=> loading information:", msg)
        print('successful load pretrained backbone from {}'.format(path))
