"""NAS cell primitives on the HIP kernels -- drop-in for the reference's `models/operations.py`.

Same public surface: `OPS[name](C, stride, affine) -> nn.Module`, same class names, constructor
arguments, attribute names and therefore the same `state_dict()` keys (`net.1.weight`,
`net.2.running_mean`, `conv1.bias`, ...).  The child `nn.Conv2d` / `nn.BatchNorm2d` modules are kept
as *parameter holders* (so `nn.SyncBatchNorm.convert_sync_batchnorm`, `.cuda()`, DDP and checkpoints
work unchanged, augment_lip_sync.py:191-208) but their own `forward` is never used: every module's
forward calls the fused HIP ops in `_ops.py`:

    ReLU is folded into the conv's load, the BatchNorm batch statistics into the conv's epilogue,
    and BN-apply is deferred (`pending()`) so the consuming cell can fuse it with the branch add.

Reference lines are cited per class.  Inputs must be CUDA tensors; there is no CPU path.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _ops as K
from ._ops import BnSide

BN_MOMENTUM = 0.1   # operations.py:27


def _use_batch_stats(bn) -> int:
    """0 / 1 / 2 = running statistics / batch statistics / batch statistics exchanged across ranks (SyncBatchNorm)."""
    return K.stats_level(bn)


class _BnOp(nn.Module):
    """An op that ends in BatchNorm: `pending(x)` returns the raw tensor + statistics, `forward`
    materialises the normalised output."""

    def pending(self, x) -> BnSide:
        raise NotImplementedError

    def forward(self, x, out=None):
        """out: K.ConcatBuffer slot -- the normalised output is written into its channel slice of a concatenation."""
        side = self.pending(x)
        return K.bn_add(side, None, relu=False, training=side.bn.training, out=out)


def pending_of(op, x) -> BnSide:
    """Deferred-BN view of any op (plain ops / containers just run)."""
    if isinstance(op, _BnOp):
        return op.pending(x)
    return BnSide(op(x))


def _is_trivial(op) -> bool:
    return isinstance(op, (Identity, Zero))


def fused_sum(op1, h1, op2, h2, out=None):
    """`op1(h1) + op2(h2)` (model_augment.py:54-59) with both BN-applies folded into the add.  out: ConcatBuffer slot."""
    helper = K.helper_stream_for_edge() if not (_is_trivial(op1) or _is_trivial(op2)) else None
    if helper is None:
        a, b = pending_of(op1, h1), pending_of(op2, h2)
    else:
        # the two edges of a node are independent: the second one runs on this stream's helper stream and joins before
        # the fused add (autograd replays the same fork in backward)
        cur = torch.cuda.current_stream()
        helper.wait_stream(cur)
        # h2 was allocated on another stream and is read on `helper` now AND by this edge's backward (the conv's weight / data
        # gradient read their saved input there): without this the allocator hands h2's block to the next allocation of its own
        # stream the moment the last reference goes, while the helper's kernels may not have run yet -- in a captured step that
        # reuse is baked in (found as a wild weight gradient of the 3x3-map decoder edge: the block had become an f64 scratch)
        K.note_stream_use(h2, helper)
        with torch.cuda.stream(helper):
            b = pending_of(op2, h2)
        a = pending_of(op1, h1)
        cur.wait_stream(helper)
        b.x.record_stream(cur)
        if b.stats is not None:
            b.stats.record_stream(cur)
    if a.bn is None and b.bn is not None:
        a, b = b, a
    training = a.bn.training if a.bn is not None else False
    return K.bn_add(a, b, relu=False, training=training, out=out)


def fused_sum_pending(op1, h1, op2, h2):
    """First half of fused_sum (current stream only): the two edges' raw outputs + statistics."""
    a, b = pending_of(op1, h1), pending_of(op2, h2)
    if a.bn is None and b.bn is not None:
        a, b = b, a
    return a, b


def fused_sum_apply(a, b, out=None):
    """Second half of fused_sum: both BatchNorm applies folded into the add."""
    training = a.bn.training if a.bn is not None else False
    return K.bn_add(a, b, relu=False, training=training, out=out)


def fused_sum_stages(op1, h1, op2, h2, result, out=None):
    """fused_sum as a generator pausing after each edge (see model_augment.Cell.stages); result: 1-element list."""
    a = pending_of(op1, h1)
    if not _is_trivial(op1):
        yield
    b = pending_of(op2, h2)
    if not _is_trivial(op2):
        yield
    if a.bn is None and b.bn is not None:
        a, b = b, a
    training = a.bn.training if a.bn is not None else False
    result[0] = K.bn_add(a, b, relu=False, training=training, out=out)


class Zero(nn.Module):
    """operations.py:31-41"""

    def __init__(self, stride):
        super().__init__()
        self.stride = stride

    def forward(self, x):
        n, c, h, w = x.shape
        if self.stride != 1:
            h, w = (h + self.stride - 1) // self.stride, (w + self.stride - 1) // self.stride
        return K.new_nhwc(n, c, h, w, x.dtype, x.device, zero=True)


class PoolBN(_BnOp):
    """AvgPool or MaxPool - BN, operations.py:44-66."""

    def __init__(self, pool_type, C, kernel_size, stride, padding, affine=True):
        super().__init__()
        pt = pool_type.lower()
        if pt == 'max':
            self.pool = nn.MaxPool2d(kernel_size, stride, padding)
        elif pt == 'avg':
            self.pool = nn.AvgPool2d(kernel_size, stride, padding, count_include_pad=False)
        else:
            raise ValueError()
        if kernel_size != 3 or padding != 1:
            raise NotImplementedError("PoolBN: only the 3x3 / pad 1 form used by OPS is implemented")
        self._avg = pt == 'avg'
        self._stride = stride
        self.bn = nn.BatchNorm2d(C, affine=affine, momentum=BN_MOMENTUM)

    def pending(self, x):
        y, st = K.pool3x3(x, self._avg, self._stride, want_stats=_use_batch_stats(self.bn))
        return BnSide(y, self.bn, st)


class ReLUConvBN(_BnOp):
    """ReLU - Conv - BN, operations.py:69-82."""

    def __init__(self, C_in, C_out, kernel_size, stride, padding, affine=True):
        super().__init__()
        self.net = nn.Sequential(
            nn.ReLU(),
            nn.Conv2d(C_in, C_out, kernel_size, stride, padding, bias=False),
            nn.BatchNorm2d(C_out, affine=affine, momentum=BN_MOMENTUM))

    _wide = None      # (WideEdges, index): this edge shares its input and geometry with other edges of its cell (see WideEdges)

    def _cb(self):
        return self.net[1], self.net[2]

    def pending(self, x):
        if self._wide is not None and K.WIDE:
            side = self._wide[0].pending(self._wide[1], x)
            if side is not None:
                return side
        conv, bn = self.net[1], self.net[2]
        y, st = K.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, relu_in=True,
                         want_stats=_use_batch_stats(bn))
        return BnSide(y, bn, st)


class WideEdges(K.WideGroup):
    """The ReLUConvBN edges of one cell that apply the same conv geometry to the same state (genotypes.py:30-54: e.g. the three
    `std_conv_3x3` on state 0 of ENCODER.normal): the first of them to be asked runs ONE conv C -> m C for all (K.conv2d_wide) and the
    others pick up their channel slice.  Anything that does not fit -- a different input object, SyncBatchNorm statistics, mixed
    train / eval BatchNorms -- returns None and the edge runs on its own."""

    def __init__(self, ops, separate_fwd=False, cross_stream=False):
        super().__init__([op._cb()[0] for op in ops], separate_fwd=separate_fwd)
        self.ops = list(ops)
        self.cross_stream = bool(cross_stream)      # its edges run on the two task branches' streams (refinement cells)
        for k, op in enumerate(self.ops):
            op._wide = (self, k)

    @staticmethod
    def mergeable(op) -> bool:
        if type(op) not in (ReLUConvBN, StdConv):
            return False
        conv = op.net[1]
        k, c = conv.kernel_size[0], conv.out_channels
        # (channel counts the LDS-DMA kernels take for C -> m C and m C -> C: multiples of 64, or 32 for the 3x3 of conv_c32)
        ci = conv.in_channels
        return (conv.stride == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
                and conv.kernel_size[0] == conv.kernel_size[1] and (ci == c or ci % 64 == 0) and c % 32 == 0)

    @staticmethod
    def fits(ops) -> bool:
        """m mergeable edges of one geometry whose merged widths the LDS-DMA kernels take: m C a multiple of 64 (conv_g4 / conv_h3 /
        conv_g8 tiles, 64-channel K-tiles of the data gradient), or 32 -> 32 3x3 (conv_c32's output / input groups)."""
        if len(ops) < 2 or not all(WideEdges.mergeable(op) for op in ops):
            return False
        w0 = ops[0].net[1].weight
        if any(op.net[1].weight.shape != w0.shape or op.net[1].padding != ops[0].net[1].padding for op in ops):
            return False
        c, ci, k = w0.shape[0], w0.shape[1], w0.shape[2]
        return (len(ops) * c) % 64 == 0 or (c == 32 and ci == 32 and k == 3 and len(ops) <= 4)

    @staticmethod
    def fits_mixed(ops) -> bool:
        """Consumers of one tensor with one kernel geometry but any widths / biases (ops with a _cb() accessor): the forward convs that
        cannot share a launch keep their own, the data gradient is merged -- its K = sum(Cout) must be whole 64-channel tiles."""
        convs = [op._cb()[0] for op in ops]
        c0 = convs[0]
        if len(ops) < 2 or any(c.kernel_size != c0.kernel_size or c.padding != c0.padding or c.in_channels != c0.in_channels
                               or c.stride != (1, 1) or c.dilation != (1, 1) or c.groups != 1 for c in convs):
            return False
        return c0.kernel_size[0] == c0.kernel_size[1] and c0.in_channels % 64 == 0 and sum(c.out_channels for c in convs) % 64 == 0 \
            and all(c.out_channels % 32 == 0 for c in convs)

    def pending(self, k, x):
        st = self.calls
        if st is None or st[0] is not x:
            if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4):
                return None
            bns = [op._cb()[1] for op in self.ops]
            want = {_use_batch_stats(bn) for bn in bns}
            if len(want) != 1 or len({bn.training for bn in bns}) != 1:
                return None
            if 2 in want and (len({id(K._sync_group(bn)[0]) for bn in bns}) != 1 or not K.WIDE_SYNC
                              or (self.cross_stream and not K.fold_ready(K._sync_group(bns[0])[0]))):
                return None      # (SyncBatchNorm over different process groups, NPP_WIDE_SYNC=0, or edges that are picked up on two
                                 #  streams while the statistics exchange is a launch ordered on the stream of the pool that holds
                                 #  them; with the exchange inside the consuming kernels -- K.fold_ready -- every edge trades its
                                 #  own slice on its own stream's channel)
            conv = self.convs[0]
            lvl = want.pop()
            ys, svs, scs, slots = K.conv2d_wide(x, self, conv.padding, True, lvl, bias_dead=lvl > 0)      # (every member is conv -> BN)
            lead = {}       # run (statistics row) -> its first edge: under SyncBatchNorm that edge's segment carries the whole row
            sides = []
            for i in range(len(bns)):
                run = next(r for r, members in enumerate(self.runs) if i in members)
                sides.append(BnSide(ys[i], bns[i], svs[i], stats_c=scs[i], gslot=slots[i], rider=run in lead))
                lead.setdefault(run, i)
            for members in self.runs:      # (K._fold_forward: who carries a run's rows once its lead edge's exchange is folded into a kernel)
                mates = [sides[i] for i in members]
                for sd in mates:
                    sd.mates = mates
            cur = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(cur)
            self.calls = st = [x, sides, cur, ev]
        side, st[1][k] = st[1][k], None
        if side is not None:
            # an edge that is picked up on ANOTHER stream than the one the merged conv ran on (the refinement cells of the two task
            # branches read the same two tensors, model_augment.py:555-571): that stream waits for the conv and owns the tensors too
            cur = torch.cuda.current_stream()
            if cur.cuda_stream != st[2].cuda_stream:
                cur.wait_event(st[3])
                side.x.record_stream(cur)
                if side.stats is not None:
                    side.stats.record_stream(cur)
        if all(sd is None for sd in st[1]):
            self.calls = None
        return side


def group_wide_edges(ops, names, indices, strides):
    """WideEdges for every set of >= 2 mergeable edges of a cell with the same (primitive, input state); `ops` may be wrapped in an
    nn.Sequential(op, Interpolate) (decoder / fusion cells)."""
    by = {}
    for op, name, idx, stride in zip(ops, names, indices, strides):
        base = op[0] if isinstance(op, nn.Sequential) else op
        if stride == 1 and WideEdges.mergeable(base):
            by.setdefault((name, idx), []).append(base)
    return [WideEdges(v) for v in by.values() if WideEdges.fits(v)]


class DilConv(_BnOp):
    """ReLU - dense dilated conv - BN, operations.py:85-101 (unused by the fixed genotype)."""

    def __init__(self, C_in, C_out, kernel_size, stride, padding, dilation, affine=True):
        super().__init__()
        self.net = nn.Sequential(
            nn.ReLU(),
            nn.Conv2d(C_in, C_out, kernel_size, stride, padding, dilation=dilation, bias=False),
            nn.BatchNorm2d(C_out, affine=affine, momentum=BN_MOMENTUM))

    pending = ReLUConvBN.pending


class StdConv(ReLUConvBN):
    """operations.py:159-172 (same computation as ReLUConvBN)."""


class SE_Block(nn.Module):
    """Squeeze-excite gate, operations.py:105-129.  `bn`/`pool2` exist (state-dict keys, never-used
    parameters when stride == 1) exactly as in the reference."""

    def __init__(self, C_in, stride, affine=True):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.conv1 = nn.Conv2d(C_in, C_in // 2, 1, 1, 0)
        self.conv2 = nn.Conv2d(C_in // 2, C_in, 1, 1, 0)
        self.relu = nn.ReLU()
        self.stride = stride
        self.pool2 = nn.AvgPool2d(2)
        self.bn = nn.BatchNorm2d(C_in, momentum=BN_MOMENTUM)

    _pair = None      # (SEPair, index): the other `se_connect` edge of the cell reads the same state (see SEPair)

    def _gate(self, x):
        if self._pair is not None and K.SE_PAIR and K.SE_FUSED:
            y = self._pair[0].gate(self._pair[1], x)
            if y is not None:
                return y
        return K.se_scale(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias)

    def pending(self, x):
        out = self._gate(x)
        if self.stride == 1:
            return BnSide(out, private=True)
        y, st = K.pool2x2(out, True, want_stats=_use_batch_stats(self.bn))
        return BnSide(y, self.bn, st)

    def forward(self, x):
        side = self.pending(x)
        if side.bn is None:
            return side.x
        return K.bn_add(side, None, relu=False, training=self.bn.training)


class SEPair:
    """The two `se_connect` edges an encoder cell applies to ONE state (ENCODER.normal / .reduce, genotypes.py:30-36): the first of
    them to be asked computes both gates in one launch pair (K.se_scale_pair: one squeeze pass, one gate-and-scale launch; backward one
    partial-sum launch and one apply launch that sums the two gradients), the second picks up its result."""

    def __init__(self, a, b):
        self.ops = [a, b]
        self.calls = None
        a._pair, b._pair = (self, 0), (self, 1)

    def gate(self, k, x):
        st = self.calls
        if st is None or st[0] is not x:
            if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4) or not K.lib().npp_se_supported(x.shape[1]):
                return None
            ys = K.se_scale_pair(x, [(m.conv1.weight, m.conv1.bias, m.conv2.weight, m.conv2.bias) for m in self.ops])
            self.calls = st = [x, list(ys)]
        y, st[1][k] = st[1][k], None
        if all(v is None for v in st[1]):
            self.calls = None
        return y


def group_se_pairs(ops, names, indices):
    """SEPair for every two SE_Block edges of a cell on the same state (the op may be wrapped in nn.Sequential(op, Interpolate))."""
    by = {}
    for op, name, idx in zip(ops, names, indices):
        base = op[0] if isinstance(op, nn.Sequential) else op
        if isinstance(base, SE_Block):
            by.setdefault(idx, []).append(base)
    return [SEPair(v[0], v[1]) for v in by.values() if len(v) == 2 and v[0].conv1.weight.shape == v[1].conv1.weight.shape]


class Identity(nn.Module):
    """operations.py:133-138"""

    def __init__(self):
        super().__init__()

    def forward(self, x):
        return x


class FactorizedReduce(_BnOp):
    """Stride-2 channel-split pointwise reduce, operations.py:142-157.  The `x[:, :, 1:, 1:]` view of the
    second conv is a padding of -1 in the kernel's geometry: no sliced copy is made."""

    def __init__(self, C_in, C_out, affine=True):
        super().__init__()
        self.relu = nn.ReLU()
        self.conv1 = nn.Conv2d(C_in, C_out // 2, 1, stride=2, padding=0, bias=False)
        self.conv2 = nn.Conv2d(C_in, C_out // 2, 1, stride=2, padding=0, bias=False)
        self.bn = nn.BatchNorm2d(C_out, affine=affine, momentum=BN_MOMENTUM)

    def pending(self, x):
        want = _use_batch_stats(self.bn)
        want = min(want, 1)     # the halves are stitched below, outside the SyncBN exchange pool
        y1, s1 = K.conv2d(x, self.conv1.weight, None, 2, 0, 1, relu_in=True, want_stats=want)
        y2, s2 = K.conv2d_crop(x, self.conv2.weight, 2, relu_in=True, want_stats=want)
        if y1.shape[2:] != y2.shape[2:]:
            raise RuntimeError("FactorizedReduce needs even spatial extents (torch.cat would fail too)")
        y = K.concat([y1, y2])
        st = None
        if want:
            c = y1.shape[1]
            a, b = s1.view(-1, 2 * c), s2.view(-1, 2 * c)      # [replica][sum | sumsq] per half
            st = torch.cat([a[:, :c], b[:, :c], a[:, c:], b[:, c:]], dim=1).reshape(-1)
        return BnSide(y, self.bn, st)


class FacConv(_BnOp):
    """ReLU - Conv(Kx1) - Conv(1xK) - BN, operations.py:174-188."""

    def __init__(self, C_in, C_out, kernel_length, stride, padding, affine=True):
        super().__init__()
        self.net = nn.Sequential(
            nn.ReLU(),
            nn.Conv2d(C_in, C_in, (kernel_length, 1), (stride, 1), (padding, 0), bias=False),
            nn.Conv2d(C_in, C_out, (1, kernel_length), (1, stride), (0, padding), bias=False),
            nn.BatchNorm2d(C_out, affine=affine, momentum=BN_MOMENTUM))

    def pending(self, x):
        c1, c2, bn = self.net[1], self.net[2], self.net[3]
        y, _ = K.conv2d(x, c1.weight, None, c1.stride, c1.padding, 1, relu_in=True)
        y, st = K.conv2d(y, c2.weight, None, c2.stride, c2.padding, 1, relu_in=False, want_stats=_use_batch_stats(bn),
                         private_in=True)
        return BnSide(y, bn, st)


class DilConvS(_BnOp):
    """ReLU - depthwise (dilated) - pointwise - BN, operations.py:202-220."""

    def __init__(self, C_in, C_out, kernel_size, stride, padding, dilation, affine=True):
        super().__init__()
        self.net = nn.Sequential(
            nn.ReLU(),
            nn.Conv2d(C_in, C_in, kernel_size, stride, padding, dilation=dilation, groups=C_in, bias=False),
            nn.Conv2d(C_in, C_out, 1, stride=1, padding=0, bias=False),
            nn.BatchNorm2d(C_out, affine=affine, momentum=BN_MOMENTUM))

    def pending(self, x):
        dw, pw, bn = self.net[1], self.net[2], self.net[3]
        y = K.dwconv2d(x, dw.weight, dw.stride[0], dw.padding[0], dw.dilation[0], relu_in=True)
        y, st = K.conv2d(y, pw.weight, None, 1, 0, 1, relu_in=False, want_stats=_use_batch_stats(bn), private_in=True)
        return BnSide(y, bn, st)


class Sep_Conv(_BnOp):
    """Two stacked DilConvS (dilation 1), operations.py:190-200."""

    def __init__(self, C_in, C_out, kernel_size, stride, padding, affine=True):
        super().__init__()
        self.net = nn.Sequential(
            DilConvS(C_in, C_in, kernel_size, stride, padding, dilation=1, affine=affine),
            DilConvS(C_in, C_out, kernel_size, 1, padding, dilation=1, affine=affine))

    def pending(self, x):
        return self.net[1].pending(self.net[0](x))


class Pooled_Conv(nn.Module):
    """AvgPool2 - [ReLU - Conv3x3(bias) - BN] x n - bilinear x2 (x2), operations.py:222-251."""

    def __init__(self, C_in, C_out, kernel_size, stride, padding, conv_nums, affine=True):
        super().__init__()
        layers = [nn.AvgPool2d(2, 2)]
        for _ in range(conv_nums):
            layers.append(nn.ReLU())
            layers.append(nn.Conv2d(C_in, C_out, kernel_size, stride, padding))
            layers.append(nn.BatchNorm2d(C_out, affine=affine, momentum=BN_MOMENTUM))
        layers.append(nn.UpsamplingBilinear2d(scale_factor=2))
        if conv_nums == 2 and stride == 2:
            layers.append(nn.UpsamplingBilinear2d(scale_factor=2))
        self.net = nn.Sequential(*layers)
        self._n = conv_nums
        self._ups = 2 if (conv_nums == 2 and stride == 2) else 1

    def forward(self, x):
        y, _ = K.pool2x2(x, True)
        for i in range(self._n):
            conv, bn = self.net[2 + 3 * i], self.net[3 + 3 * i]
            y, st = K.conv2d(y, conv.weight, conv.bias, conv.stride, conv.padding, 1, relu_in=True,
                             want_stats=_use_batch_stats(bn), bias_dead=_use_batch_stats(bn) > 0)
            y = K.bn_add(BnSide(y, bn, st), None, relu=False, training=bn.training)
        for _ in range(self._ups):
            y = K.bilinear(y, y.shape[2] * 2, y.shape[3] * 2)
        return y


OPS = {   # operations.py:9-25
    'none': lambda C, stride, affine: Zero(stride),
    'avg_pool_3x3': lambda C, stride, affine: PoolBN('avg', C, 3, stride, 1, affine=affine),
    'max_pool_3x3': lambda C, stride, affine: PoolBN('max', C, 3, stride, 1, affine=affine),
    'skip_connect': lambda C, stride, affine: Identity() if stride == 1 else FactorizedReduce(C, C, affine=affine),
    'std_conv_3x3': lambda C, stride, affine: ReLUConvBN(C, C, 3, stride, 1, affine=affine),
    'std_conv_1x1': lambda C, stride, affine: ReLUConvBN(C, C, 1, stride, 0, affine=affine),
    'dil_conv_3x3_2': lambda C, stride, affine: DilConvS(C, C, 3, stride, 2, 2, affine=affine),
    'dil_conv_3x3_4': lambda C, stride, affine: DilConvS(C, C, 3, stride, 4, 4, affine=affine),
    'dil_conv_5x5_4': lambda C, stride, affine: DilConvS(C, C, 5, stride, 4, 2, affine=affine),
    'se_connect': lambda C, stride, affine: SE_Block(C, stride, affine=affine),
    'conv_7x1_1x7': lambda C, stride, affine: FacConv(C, C, 7, stride, 3, affine=affine),
    'sep_conv_3x3': lambda C, stride, affine: Sep_Conv(C, C, 3, stride, 1, affine=affine),
    'sep_conv_5x5': lambda C, stride, affine: Sep_Conv(C, C, 5, stride, 2, affine=affine),
    'poled_conv_x1': lambda C, stride, affine: Pooled_Conv(C, C, 3, stride, 1, 1, affine=affine),
    'poled_conv_x2': lambda C, stride, affine: Pooled_Conv(C, C, 3, stride, 1, 2, affine=affine),
}
