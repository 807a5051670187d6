"""CPU oracle for the NPPNet hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this file.  The product path (`npp_amd/`) never does; it fails loudly when the HIP
library is missing.

What this is: a from-scratch *functional* restatement (state-dict keyed, no nn.Module
tree) of the reference's fixed-genotype network and its two criteria, evaluated with
PyTorch fp32/fp64 CPU arithmetic -- the same ATen-CPU arithmetic the reference itself
runs on a GPU-less host.  Each function cites the reference lines it follows
(paths relative to /root/reference).

Pinning: the reference has no tests and no golden vectors, so this
oracle is pinned against outputs of the reference itself, generated in the build
container by `oracle/make_golden.py` (which imports /root/reference) and committed
under `tests/golden/`; `tests/test_oracle_golden.py` checks the oracle against them.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

BN_MOMENTUM = 0.1  # models/operations.py:27
BN_EPS = 1e-5      # nn.BatchNorm2d default

# ---- genotype constants (models/genotypes.py:30-54), restated as data -------------------
ENC_NORMAL = [('std_conv_3x3', 0), ('se_connect', 1), ('se_connect', 1), ('std_conv_3x3', 0),
              ('max_pool_3x3', 1), ('std_conv_3x3', 2), ('std_conv_3x3', 3), ('std_conv_3x3', 0)]
ENC_REDUCE = [('std_conv_3x3', 0), ('se_connect', 1), ('se_connect', 1), ('std_conv_3x3', 2),
              ('dil_conv_3x3_4', 3), ('dil_conv_3x3_4', 2), ('max_pool_3x3', 3), ('dil_conv_3x3_2', 0)]
DEC_UP1 = [('std_conv_1x1', 1), ('std_conv_1x1', 0), ('std_conv_1x1', 1), ('std_conv_3x3', 0),
           ('std_conv_1x1', 0), ('dil_conv_3x3_2', 1), ('std_conv_3x3', 3), ('std_conv_1x1', 1)]
DEC_UP2 = [('std_conv_3x3', 1), ('se_connect', 0), ('dil_conv_3x3_2', 2), ('std_conv_1x1', 1),
           ('poled_conv_x1', 3), ('std_conv_1x1', 2), ('std_conv_3x3', 1), ('std_conv_1x1', 2)]
INTER_T1 = [[('dil_conv_3x3_2', 0)], [('std_conv_3x3', 1)], [('std_conv_1x1', 1), ('std_conv_3x3', 2)],
            [('std_conv_1x1', 2), ('std_conv_3x3', 3)]]
INTER_T2 = [[('dil_conv_3x3_2', 0)], [('poled_conv_x1', 1)], [('std_conv_1x1', 2)],
            [('std_conv_3x3', 1), ('std_conv_3x3', 3)]]
INTER_T3 = [[('dil_conv_3x3_2', 4), ('dil_conv_3x3_2', 2), ('dil_conv_3x3_2', 1)],
            [('std_conv_3x3', 1), ('std_conv_3x3', 2), ('dil_conv_3x3_2', 5), ('dil_conv_3x3_2', 0)],
            [('std_conv_3x3', 1), ('dil_conv_3x3_2', 2), ('dil_conv_3x3_4', 5), ('dil_conv_3x3_2', 3)]]
INTER_T4 = [[('std_conv_3x3', 0)], [('std_conv_3x3', 1)], [('std_conv_1x1', 2), ('std_conv_3x3', 1)]]
FUSE_POSE = [('std_conv_3x3', 1), ('std_conv_3x3', 2), ('std_conv_3x3', 0), ('max_pool_3x3', 2),
             ('std_conv_3x3', 4), ('std_conv_3x3', 2), ('std_conv_3x3', 4), ('std_conv_3x3', 3)]
FUSE_PAR = [('dil_conv_3x3_2', 2), ('se_connect', 1), ('dil_conv_3x3_2', 2), ('dil_conv_3x3_2', 3),
            ('max_pool_3x3', 3), ('std_conv_3x3', 2), ('dil_conv_3x3_2', 5), ('std_conv_3x3', 2)]

# LIP class weights, core/criterion.py:17-21
WEIGHTS_LIP = [0.7602572, 0.94236198, 0.85644457, 1.04346266, 1.10627293, 0.80980162,
               0.95168713, 0.8403769, 1.05798412, 0.85746254, 1.01274366, 1.05854692,
               1.03430773, 0.84867818, 0.88027721, 0.87580925, 0.98747462, 0.9876475,
               1.00016535, 1.00108882]


class Ctx:
    """Parameters/buffers by state-dict key + the train flag + buffer updates collected on the way."""

    def __init__(self, tensors: Dict[str, torch.Tensor], train: bool):
        self.t = tensors
        self.train = train
        self.new_buffers: Dict[str, torch.Tensor] = {}

    def __getitem__(self, k):
        return self.t[k]

    def has(self, k):
        return k in self.t


# ---- primitives (models/operations.py) ---------------------------------------------------
def bn(c: Ctx, pre: str, x: torch.Tensor, affine: bool = True) -> torch.Tensor:
    """nn.BatchNorm2d(momentum=0.1, eps=1e-5) -- operations.py:61,78,216,241; SURVEY §8 a20.
    Train: batch mean / biased var normalise; running stats get the unbiased var."""
    w = c[pre + 'weight'] if affine and c.has(pre + 'weight') else None
    b = c[pre + 'bias'] if affine and c.has(pre + 'bias') else None
    rm, rv = c[pre + 'running_mean'], c[pre + 'running_var']
    if c.train:
        xd = x.detach()
        n = xd.numel() // xd.shape[1]
        mean = xd.mean(dim=(0, 2, 3))
        var_b = xd.var(dim=(0, 2, 3), unbiased=False)
        c.new_buffers[pre + 'running_mean'] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
        c.new_buffers[pre + 'running_var'] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var_b * (n / max(n - 1, 1))
        return F.batch_norm(x, None, None, w, b, True, BN_MOMENTUM, BN_EPS)
    return F.batch_norm(x, rm, rv, w, b, False, BN_MOMENTUM, BN_EPS)


def relu_conv_bn(c, pre, x, k, stride, pad):
    """ReLUConvBN, operations.py:69-82."""
    y = F.conv2d(F.relu(x), c[pre + 'net.1.weight'], None, stride, pad)
    return bn(c, pre + 'net.2.', y)


def dil_conv_s(c, pre, x, k, stride, pad, dil):
    """DilConvS, operations.py:202-220: ReLU - depthwise(dil) - pointwise - BN."""
    C = x.shape[1]
    y = F.conv2d(F.relu(x), c[pre + 'net.1.weight'], None, stride, pad, dil, groups=C)
    y = F.conv2d(y, c[pre + 'net.2.weight'])
    return bn(c, pre + 'net.3.', y)


def sep_conv(c, pre, x, k, stride, pad):
    """Sep_Conv, operations.py:190-200."""
    y = dil_conv_s(c, pre + 'net.0.', x, k, stride, pad, 1)
    return dil_conv_s(c, pre + 'net.1.', y, k, 1, pad, 1)


def pool_bn(c, pre, x, kind, stride):
    """PoolBN, operations.py:44-66 (avg: count_include_pad=False)."""
    if kind == 'max':
        y = F.max_pool2d(x, 3, stride, 1)
    else:
        y = F.avg_pool2d(x, 3, stride, 1, count_include_pad=False)
    return bn(c, pre + 'bn.', y)


def se_block(c, pre, x, stride):
    """SE_Block, operations.py:105-129 (bn/pool2 only used when stride == 2)."""
    w = F.adaptive_avg_pool2d(x, 1)
    w = F.relu(F.conv2d(w, c[pre + 'conv1.weight'], c[pre + 'conv1.bias']))
    w = torch.sigmoid(F.conv2d(w, c[pre + 'conv2.weight'], c[pre + 'conv2.bias']))
    out = x * w
    if stride == 1:
        return out
    return bn(c, pre + 'bn.', F.avg_pool2d(out, 2))


def factorized_reduce(c, pre, x):
    """FactorizedReduce, operations.py:142-157."""
    x = F.relu(x)
    y = torch.cat([F.conv2d(x, c[pre + 'conv1.weight'], None, 2),
                   F.conv2d(x[:, :, 1:, 1:], c[pre + 'conv2.weight'], None, 2)], dim=1)
    return bn(c, pre + 'bn.', y)


def fac_conv(c, pre, x, k, stride, pad):
    """FacConv, operations.py:174-188."""
    y = F.conv2d(F.relu(x), c[pre + 'net.1.weight'], None, (stride, 1), (pad, 0))
    y = F.conv2d(y, c[pre + 'net.2.weight'], None, (1, stride), (0, pad))
    return bn(c, pre + 'net.3.', y)


def interp(x, scale):
    """Interpolate, model_augment.py:109-116 (bilinear, align_corners=True)."""
    return F.interpolate(x, scale_factor=scale, mode='bilinear', align_corners=True)


def pooled_conv(c, pre, x, stride, n):
    """Pooled_Conv, operations.py:222-251: avgpool2 - [ReLU-conv3x3(bias)-BN]*n - up x2 (- up x2)."""
    y = F.avg_pool2d(x, 2, 2)
    idx = 1
    for _ in range(n):
        y = F.conv2d(F.relu(y), c[pre + f'net.{idx + 1}.weight'], c[pre + f'net.{idx + 1}.bias'], stride, 1)
        y = bn(c, pre + f'net.{idx + 2}.', y)
        idx += 3
    y = interp(y, 2)
    if n == 2 and stride == 2:
        y = interp(y, 2)
    return y


def apply_op(c, name, pre, x, stride):
    """OPS registry dispatch, operations.py:9-25."""
    if name == 'none':
        return x * 0. if stride == 1 else x[:, :, ::stride, ::stride] * 0
    if name == 'avg_pool_3x3':
        return pool_bn(c, pre, x, 'avg', stride)
    if name == 'max_pool_3x3':
        return pool_bn(c, pre, x, 'max', stride)
    if name == 'skip_connect':
        return x if stride == 1 else factorized_reduce(c, pre, x)
    if name == 'std_conv_3x3':
        return relu_conv_bn(c, pre, x, 3, stride, 1)
    if name == 'std_conv_1x1':
        return relu_conv_bn(c, pre, x, 1, stride, 0)
    if name == 'dil_conv_3x3_2':
        return dil_conv_s(c, pre, x, 3, stride, 2, 2)
    if name == 'dil_conv_3x3_4':
        return dil_conv_s(c, pre, x, 3, stride, 4, 4)
    if name == 'dil_conv_5x5_4':
        return dil_conv_s(c, pre, x, 5, stride, 4, 2)
    if name == 'se_connect':
        return se_block(c, pre, x, stride)
    if name == 'conv_7x1_1x7':
        return fac_conv(c, pre, x, 7, stride, 3)
    if name == 'sep_conv_3x3':
        return sep_conv(c, pre, x, 3, stride, 1)
    if name == 'sep_conv_5x5':
        return sep_conv(c, pre, x, 5, stride, 2)
    if name == 'poled_conv_x1':
        return pooled_conv(c, pre, x, stride, 1)
    if name == 'poled_conv_x2':
        return pooled_conv(c, pre, x, stride, 2)
    raise KeyError(name)


# ---- cells (models/model_augment.py:16-229) ---------------------------------------------
def _dag(c, pre, states, plan, up_on_input0, up_scales=None):
    """4 steps x 2 ops; s = op1(h1) + op2(h2)  (model_augment.py:48-62, 92-106, 153-172).
    up_scales: {input index: scale} -- nn.Sequential(op, Interpolate(scale)) on the ops fed by those inputs (order == 0 fuse cells,
    model_augment.py:143-147, 202-206)."""
    for i in range(len(plan) // 2):
        hs = []
        for j in (2 * i, 2 * i + 1):
            name, idx, stride = plan[j]
            if up_scales is not None and idx in up_scales:
                h = interp(apply_op(c, name, f'{pre}_ops.{j}.0.', states[idx], stride), up_scales[idx])
            elif up_on_input0 and idx == 0:   # nn.Sequential(op, Interpolate(2)), model_augment.py:86-87
                h = interp(apply_op(c, name, f'{pre}_ops.{j}.0.', states[idx], stride), 2)
            else:
                h = apply_op(c, name, f'{pre}_ops.{j}.', states[idx], stride)
            hs.append(h)
        states.append(hs[0] + hs[1])
    return states


def cell(c, pre, s0, s1, reduction, reduction_prev):
    """Cell, model_augment.py:16-62."""
    s0 = factorized_reduce(c, pre + 'preprocess0.', s0) if reduction_prev \
        else relu_conv_bn(c, pre + 'preprocess0.', s0, 1, 1, 0)
    s1 = relu_conv_bn(c, pre + 'preprocess1.', s1, 1, 1, 0)
    geno = ENC_REDUCE if reduction else ENC_NORMAL
    plan = [(n, i, 2 if reduction and i < 2 else 1) for n, i in geno]
    st = _dag(c, pre, [s0, s1], plan, False)
    return torch.cat(st[2:6], dim=1)


def upsample_cell(c, pre, s0, s1, geno):
    """Upsample, model_augment.py:64-106."""
    s0 = relu_conv_bn(c, pre + 'preprocess0.', s0, 1, 1, 0)
    s1 = relu_conv_bn(c, pre + 'preprocess1.', s1, 1, 1, 0)
    st = _dag(c, pre, [s0, s1], [(n, i, 1) for n, i in geno], True)
    return torch.cat(st[2:6], dim=1)


def fuse_cell(c, pre, s0, s1, s2, geno):
    """PoseCell1 / ParCell1 with order=1, model_augment.py:119-229."""
    s0 = relu_conv_bn(c, pre + 'preprocess0.', s0, 1, 1, 0)
    s1 = relu_conv_bn(c, pre + 'preprocess1.', s1, 1, 1, 0)
    s2 = relu_conv_bn(c, pre + 'preprocess2.', s2, 1, 1, 0)
    st = _dag(c, pre, [s0, s1, s2], [(n, i, 1) for n, i in geno], False)
    return torch.cat(st[0:3], dim=1), torch.cat(st[3:7], dim=1)


def fuse_cell_order0(c, pre, s0, s1, s2, geno):
    """PoseCell1 / ParCell1 with order=0, model_augment.py:119-172 / 174-229: inputs at 1/4, 1/2 and full resolution; the ops on inputs
    0 / 1 are followed by Interpolate(4) / Interpolate(2) (bilinear, align_corners=True, :143-147); after the node loop states 0 and 1
    are replaced by F.interpolate(scale_factor=4 / 2) in its default nearest mode (:167-169) before both concatenations."""
    s0 = relu_conv_bn(c, pre + 'preprocess0.', s0, 1, 1, 0)
    s1 = relu_conv_bn(c, pre + 'preprocess1.', s1, 1, 1, 0)
    s2 = relu_conv_bn(c, pre + 'preprocess2.', s2, 1, 1, 0)
    st = _dag(c, pre, [s0, s1, s2], [(n, i, 1) for n, i in geno], False, up_scales={0: 4, 1: 2})
    st[0] = F.interpolate(st[0], scale_factor=4)
    st[1] = F.interpolate(st[1], scale_factor=2)
    return torch.cat(st[0:3], dim=1), torch.cat(st[3:7], dim=1)


def cell_case(c, pre, spec, xs):
    """One entry of oracle/cases.py CELL_CASES on the inputs `xs` (None where the case has no such input); returns the list
    of outputs.  `pre` is the state-dict prefix of the block."""
    kind = spec["kind"]
    if kind == "cell":
        a = spec["args"]
        return [cell(c, pre, xs[0], xs[1], a[3], a[4])]
    if kind == "upsample":
        return [upsample_cell(c, pre, xs[0], xs[1], DEC_UP1 if spec["which"] == 1 else DEC_UP2)]
    if kind in ("pose", "par"):
        fn = fuse_cell_order0 if spec["args"][3] == 0 else fuse_cell
        return list(fn(c, pre, xs[0], xs[1], xs[2], FUSE_POSE if kind == "pose" else FUSE_PAR))
    geno = {1: INTER_T1, 2: INTER_T2, 3: INTER_T3, 4: INTER_T4}[spec["task"]][spec["stage"]]
    st = spec["stage"]
    res = [1, 1 / 2, 1 / 4, 1 / 8, 1 / 4, 1 / 2, 1]
    z = 0
    for j, (name, ind) in enumerate(geno):
        if kind == "inter":            # Network._compile, model_augment.py:576-599
            z = z + _inter_op(c, f'{pre}{j}.', name, xs[ind], ind != st, 1 / 2 ** (st - ind))
        else:                          # Network._compile3, model_augment.py:626-649
            z = z + _inter_op(c, f'{pre}{j}.', name, xs[ind], ind != 4 + st, res[4 + st] / res[ind])
    return [z]


# ---- network (models/model_augment.py:231-574) -------------------------------------------
def _stem(c, pre, x, stride, relu):
    y = bn(c, pre + '1.', F.conv2d(x, c[pre + '0.weight'], None, stride, 1))
    return F.relu(y) if relu else y


def _head2(c, pre, x, k):
    """ReLU - conv(k, bias?) - BN - ReLU - conv1x1(bias), model_augment.py:365-398."""
    b = c[pre + '1.bias'] if c.has(pre + '1.bias') else None
    y = F.conv2d(F.relu(x), c[pre + '1.weight'], b, 1, k // 2)
    y = F.relu(bn(c, pre + '2.', y))
    return F.conv2d(y, c[pre + '4.weight'], c[pre + '4.bias'])


def _layer(c, pre, x):
    """ReLU - conv1x1(bias) - BN, model_augment.py:332-351."""
    return bn(c, pre + '2.', F.conv2d(F.relu(x), c[pre + '1.weight'], c[pre + '1.bias']))


def _inter_op(c, pre, name, x, has_extra, scale):
    """One cross-task edge built by _compile/_compile3 (model_augment.py:576-599, 626-649):
    OPS[name](C_src) [+ Interpolate(scale) + Conv2d 1x1 (bias)]."""
    if not has_extra:
        return apply_op(c, name, pre, x, 1)
    y = apply_op(c, name, pre + '0.', x, 1)
    y = interp(y, scale)
    return F.conv2d(y, c[pre + '1.1.weight'], c[pre + '1.1.bias'])


def network_forward(tensors: Dict[str, torch.Tensor], x: torch.Tensor, layers: int = 16,
                    refine_layers: int = 1, train: bool = True):
    """Network.forward, model_augment.py:402-574.  Returns (pose_list, par_list, new_buffers)."""
    c = Ctx(tensors, train)
    L = layers
    s0 = _stem(c, 'stem1.', _stem(c, 'stem0.', x, 2, True), 2, True)
    s1 = _stem(c, 'stem2.', s0, 1, False)
    s2 = _stem(c, 'stem4.', _stem(c, 'stem3.', x, 2, True), 2, True)
    s3 = _stem(c, 'stem5.', s2, 1, False)
    f1: List[torch.Tensor] = []
    f2: List[torch.Tensor] = []
    taps = [L // 4 - 1, 2 * L // 4 - 1, 3 * L // 4 - 1, L - 1]
    reds = [L // 4, 2 * L // 4, 3 * L // 4]
    red_prev = False
    k1 = k2 = 0
    for i in range(L):
        red = i in reds
        s0, s1 = s1, cell(c, f'cells1.{i}.', s0, s1, red, red_prev)
        s2, s3 = s3, cell(c, f'cells2.{i}.', s2, s3, red, red_prev)
        red_prev = red
        if i in taps:
            stage = taps.index(i)
            f1.append(s1)
            f2.append(s3)
            z1 = 0
            for name, ind in INTER_T1[stage]:
                z1 = z1 + _inter_op(c, f'_ops1.{k1}.', name, f2[ind], ind != stage, 1 / 2 ** (stage - ind))
                k1 += 1
            z2 = 0
            for name, ind in INTER_T2[stage]:
                z2 = z2 + _inter_op(c, f'_ops2.{k2}.', name, f1[ind], ind != stage, 1 / 2 ** (stage - ind))
                k2 += 1
            s1 = s1 + z1
            s3 = s3 + z2
            f1[-1] = s1
            f2[-1] = s3
    # decoder, model_augment.py:448-533 (three structurally identical stages)
    res = [1, 1 / 2, 1 / 4, 1 / 8, 1 / 4, 1 / 2, 1]
    k1 = k2 = 0
    for d in range(3):
        coarse1 = f1[3] if d == 0 else f1[-1]
        coarse2 = f2[3] if d == 0 else f2[-1]
        o1 = upsample_cell(c, f'upsamples1.{d}.', coarse1, f1[2 - d], DEC_UP1)
        o2 = upsample_cell(c, f'upsamples2.{d}.', coarse2, f2[2 - d], DEC_UP2)
        f1.append(o1)
        f2.append(o2)
        z1 = 0
        for name, ind in INTER_T3[d]:
            z1 = z1 + _inter_op(c, f'up_ops1.{k1}.', name, f2[ind], ind != 4 + d, res[4 + d] / res[ind])
            k1 += 1
        z2 = 0
        for name, ind in INTER_T4[d]:
            z2 = z2 + _inter_op(c, f'up_ops2.{k2}.', name, f1[ind], ind != 4 + d, res[4 + d] / res[ind])
            k2 += 1
        f1[-1] = o1 + z1
        f2[-1] = o2 + z2
    x1 = torch.cat((f1[0], f1[6], interp(f1[5], 2), interp(f1[4], 4)), dim=1)
    x2 = torch.cat((f2[0], f2[6], interp(f2[5], 2), interp(f2[4], 4)), dim=1)
    in1 = _layer(c, 'pose_auxlayer.', x1)
    in2 = _layer(c, 'edge_layer.', x2)
    in3 = _layer(c, 'pose_layer.', x1)
    in4 = _layer(c, 'par_layer.', x2)
    pose_list, par_list = [], []

    def heads(i):
        edge = _head2(c, f'edge_head.{i}.', in2, 3)
        pose_aux = _head2(c, f'pose_auxnet.{i}.', in1, 3)
        pose_map = _head2(c, f'pose_head.{i}.', in3, 1)
        par_map = _head2(c, f'par_head.{i}.', in4, 1)
        pose_list.append([pose_map, pose_aux])
        par_list.append([par_map, edge])

    heads(0)
    for i in range(1, refine_layers + 1):
        for j in range(3):
            m = 2 * (i - 1) + j
            n_in1, tmp = fuse_cell(c, f'pose_net.{m}.', in1, in3, in4, FUSE_POSE)
            in2, n_in4 = fuse_cell(c, f'par_net.{m}.', in2, in3, in4, FUSE_PAR)
            in1, in3, in4 = n_in1, tmp, n_in4
        heads(i)
    return pose_list, par_list, c.new_buffers


# ---- criteria (core/criterion.py) --------------------------------------------------------
def ohem_ce(score, target, thresh=0.9, min_kept=131072, ignore=255, weight=None):
    """OhemCrossEntropy.forward, core/criterion.py:54-72 (score already at label size)."""
    if weight is None:
        weight = torch.tensor(WEIGHTS_LIP, dtype=score.dtype)
    min_kept = max(1, min_kept)
    pred = F.softmax(score, dim=1)
    pixel_losses = F.cross_entropy(score, target, weight, ignore_index=ignore, reduction='none').reshape(-1)
    mask = target.reshape(-1) != ignore
    tt = target.clone()
    tt[tt == ignore] = 0
    p = pred.gather(1, tt.unsqueeze(1)).reshape(-1)[mask]
    p_sorted, ind = p.sort()
    min_value = p_sorted[min(min_kept, p_sorted.numel() - 1)]
    threshold = max(float(min_value), thresh)
    pl = pixel_losses[mask][ind]
    return pl[p_sorted < threshold].mean()


def parsing_loss(preds, target, **kw):
    """Criterion_par.parsing_loss, core/criterion.py:158-202 (preds=[par_map, edge])."""
    h, w = target[0].shape[1], target[0].shape[2]
    pos = torch.sum(target[1] == 1, dtype=torch.float)
    neg = torch.sum(target[1] == 0, dtype=torch.float)
    weights = torch.stack([pos / (pos + neg), neg / (pos + neg)]).to(preds[0].dtype)
    sp = F.interpolate(preds[0], size=(h, w), mode='bilinear', align_corners=True)
    loss = ohem_ce(sp, target[0], **kw)
    se = F.interpolate(preds[1], size=(h, w), mode='bilinear', align_corners=True)
    return loss + F.cross_entropy(se, target[1], weights, ignore_index=255)


def criterion_par(par_list, target, lamda, **kw):
    """Criterion_par.forward, core/criterion.py:204-217."""
    loss = 0.
    for i, p in enumerate(par_list):
        loss = loss + parsing_loss(p, target, **kw) * torch.exp(-lamda[i]) + lamda[i]
    return loss


def joint_loss(output, target, target_weight=None):
    """Criterion_pose.joint_loss, core/criterion.py:82-128: sum over joints of MSE(mean over batch*pixels), main + aux,
    / num_joints.  target_weight [N, J, 1] (use_target_weight=True, :103-108): prediction and target of (n, j) are both
    multiplied by it.  A map whose size differs from its target's is resampled to the MAIN target's size with
    F.interpolate(mode='bilinear') (align_corners=False), :92-96 and :113-115."""
    J = output[0].shape[1]
    h, w = target[0].shape[2:]
    loss = 0.
    for o, t in zip(output, target):
        if o.shape[2:] != t.shape[2:]:
            o = F.interpolate(o, size=(h, w), mode='bilinear')
        n = o.shape[0]
        for j in range(J):
            a, b = o[:, j].reshape(n, -1), t[:, j].reshape(n, -1)
            if target_weight is not None:
                a, b = a * target_weight[:, j], b * target_weight[:, j]
            loss = loss + F.mse_loss(a.squeeze(), b.squeeze())
    return loss / J


def criterion_pose(pose_list, target, lamda, target_weight=None):
    """Criterion_pose.forward, core/criterion.py:130-145."""
    loss = 0.
    for i, p in enumerate(pose_list):
        loss = loss + joint_loss(p, target, target_weight) * torch.exp(-lamda[i]) + lamda[i]
    return loss


def train_step_loss(tensors, images, labels_par, labels_pose, lam_pose, lam_par, layers=16,
                    refine_layers=1, train=True, **kw):
    """core/function.py:87-98: model -> criterion_par + criterion_pose -> mean."""
    pose_list, par_list, newb = network_forward(tensors, images, layers, refine_layers, train)
    lp = criterion_par(par_list, labels_par, lam_par, **kw)
    lq = criterion_pose(pose_list, labels_pose, lam_pose)
    loss = (lp.reshape(1) + lq.reshape(1)).mean()
    return loss, pose_list, par_list, newb


# ---- search supernet (models/model_search_interact.py) -------------------------------------------------------------
PRIMITIVES_INTER = ['std_conv_3x3', 'dil_conv_3x3_4', 'se_connect', 'max_pool_3x3', 'dil_conv_3x3_2', 'std_conv_1x1',
                    'poled_conv_x1']   # models/genotypes.py:20-28


def channel_shuffle2(x):
    """channel_shuffle(x, 2), model_search_interact.py:22-36."""
    n, c, h, w = x.shape
    return x.view(n, 2, c // 2, h, w).transpose(1, 2).contiguous().view(n, c, h, w)


def mixed_op(c: Ctx, pre: str, x, weights, up_scale, has_extra):
    """MixedOp.forward, model_search_interact.py:56-74 (candidates built with affine=False, :46)."""
    C = x.shape[1]
    xt, xt2 = x[:, :C // 2], x[:, C // 2:]
    temp1 = 0.
    for k, name in enumerate(PRIMITIVES_INTER):
        p = f'{pre}_ops.{k}.'
        nest = ('0.' if up_scale else '')                      # Sequential(op, Interpolate) wrapper (:50-51)
        if 'pool' in name:                                     # Sequential(PoolBN, BatchNorm2d(affine=False)) (:48-49)
            y = apply_op(c, name, p + nest + '0.', xt, 1)
            y = bn(c, p + nest + '1.', y, affine=False)
        else:
            y = apply_op(c, name, p + nest, xt, 1)
        if up_scale:
            y = interp(y, up_scale)
        temp1 = temp1 + weights[k] * y
    if up_scale:
        xt2 = F.interpolate(xt2, scale_factor=up_scale)        # default mode: nearest (:63-64)
    if temp1.shape[2] != xt2.shape[2]:
        xt2 = F.max_pool2d(xt2, 2, 2)
    ans = channel_shuffle2(torch.cat([temp1, xt2], dim=1))
    if has_extra:
        ans = F.conv2d(ans, c[pre + 'extra_conv.weight'], c[pre + 'extra_conv.bias'])
    return ans


def mixed_cell(c, pre, s0, s1, s2, weights, weights2, steps=4, multiplier=4):
    """PoseCell / ParCell with order=1, model_search_interact.py:361-378."""
    states = [relu_conv_bn(c, pre + 'preprocess0.', s0, 1, 1, 0), relu_conv_bn(c, pre + 'preprocess1.', s1, 1, 1, 0),
              relu_conv_bn(c, pre + 'preprocess2.', s2, 1, 1, 0)]
    offset = 0
    for _ in range(steps):
        s = 0.
        for j, h in enumerate(states):
            s = s + weights2[offset + j] * mixed_op(c, f'{pre}_ops.{offset + j}.', h, weights[offset + j], None, False)
        offset += len(states)
        states.append(s)
    return torch.cat(states[0:3], dim=1), torch.cat(states[-multiplier:], dim=1)


def btw(n_input, steps, betas):
    """model_search_interact.py:1054-1065."""
    parts, start, n = [], 0, n_input
    for _ in range(steps):
        parts.append(F.softmax(betas[start:start + n], dim=-1))
        start += n
        n += 1
    return torch.cat(parts, dim=0)


def search_network_forward(tensors, x, layers=16, refine_layers=1, train=True):
    """search Network.forward, model_search_interact.py:626-770."""
    c = Ctx(tensors, train)
    L = layers
    s0 = _stem(c, 'stem1.', _stem(c, 'stem0.', x, 2, True), 2, True)
    s1 = _stem(c, 'stem2.', s0, 1, False)
    s2 = _stem(c, 'stem4.', _stem(c, 'stem3.', x, 2, True), 2, True)
    s3 = _stem(c, 'stem5.', s2, 1, False)
    f1, f2 = [], []
    taps = [L // 4 - 1, 2 * L // 4 - 1, 3 * L // 4 - 1, L - 1]
    reds = [L // 4, 2 * L // 4, 3 * L // 4]
    red_prev = False
    offset = 0
    stage = 0

    def mix(name, base, feats, alphas, betas, up_of, extra_of):
        w = F.softmax(alphas, dim=-1)
        w2 = F.softmax(betas, dim=-1)
        z = 0.
        for j, h in enumerate(feats):
            z = z + w2[j] * mixed_op(c, f'{name}.{base + j}.', h, w[j], up_of(j), extra_of(j))
        return z

    for i in range(L):
        red = i in reds
        s0, s1 = s1, cell(c, f'cells1.{i}.', s0, s1, red, red_prev)
        s2, s3 = s3, cell(c, f'cells2.{i}.', s2, s3, red, red_prev)
        red_prev = red
        if i in taps:
            f1.append(s1)
            f2.append(s3)
            n = len(f1)
            up_of = lambda j, st=stage: 1 / 2 ** (st - j)          # noqa: E731
            ex_of = lambda j, st=stage: st != j                    # noqa: E731
            z1 = mix('_ops1', offset, f2, c['alphas1'][offset:offset + n], c['betas1'][offset:offset + n], up_of, ex_of)
            z2 = mix('_ops2', offset, f1, c['alphas2'][offset:offset + n], c['betas2'][offset:offset + n], up_of, ex_of)
            s1 = s1 + z1
            s3 = s3 + z2
            f1[-1], f2[-1] = s1, s3
            offset += n
            stage += 1
    res = [1, 1 / 2, 1 / 4, 1 / 8, 1 / 4, 1 / 2, 1]
    cont = 0
    for d in range(3):
        o1 = upsample_cell(c, f'upsamples1.{d}.', f1[3] if d == 0 else f1[-1], f1[2 - d], DEC_UP1)
        o2 = upsample_cell(c, f'upsamples2.{d}.', f2[3] if d == 0 else f2[-1], f2[2 - d], DEC_UP2)
        f1.append(o1)
        f2.append(o2)
        n = len(f1)
        up_of = lambda j, dd=d: res[4 + dd] / res[j]               # noqa: E731
        ex_of = lambda j, dd=d: 4 + dd != j                        # noqa: E731
        z1 = mix('up_ops1', cont, f2, c['alphas3'][cont:cont + n], c['betas3'][cont:cont + n], up_of, ex_of)
        z2 = mix('up_ops2', cont, f1, c['alphas4'][cont:cont + n], c['betas4'][cont:cont + n], up_of, ex_of)
        f1[-1], f2[-1] = o1 + z1, o2 + z2
        cont += n
    x1 = torch.cat((f1[0], f1[6], interp(f1[5], 2), interp(f1[4], 4)), dim=1)
    x2 = torch.cat((f2[0], f2[6], interp(f2[5], 2), interp(f2[4], 4)), dim=1)
    in1 = _layer(c, 'pose_auxlayer.', x1)
    in2 = _layer(c, 'edge_layer.', x2)
    in3 = _layer(c, 'pose_layer.', x1)
    in4 = _layer(c, 'par_layer.', x2)
    pose_list, par_list = [], []

    def heads(i):
        edge = _head2(c, f'edge_head.{i}.', in2, 3)
        pose_aux = _head2(c, f'pose_auxnet.{i}.', in1, 3)
        pose_map = _head2(c, f'pose_head.{i}.', in3, 1)
        par_map = _head2(c, f'par_head.{i}.', in4, 1)
        pose_list.append([pose_map, pose_aux])
        par_list.append([par_map, edge])

    heads(0)
    w_pose, w_pose2 = F.softmax(c['alphas_pose'], dim=-1), btw(3, 4, c['betas_pose'])
    w_par, w_par2 = F.softmax(c['alphas_par'], dim=-1), btw(3, 4, c['betas_par'])
    for i in range(1, refine_layers + 1):
        for j in range(3):
            m = 2 * (i - 1) + j
            n1, tmp = mixed_cell(c, f'pose_net.{m}.', in1, in3, in4, w_pose, w_pose2)
            in2, n4 = mixed_cell(c, f'par_net.{m}.', in2, in3, in4, w_par, w_par2)
            in1, in3, in4 = n1, tmp, n4
        heads(i)
    return pose_list, par_list, c.new_buffers


# ---- evaluation: flip-TTA parsing + confusion matrix --------------------------------------------------
def tta_parsing_logits(pred_par, flip_pred_par, size):
    """core/function.py:925-943 (`validate_sync`): both logit maps are upsampled with
    `F.interpolate(mode='bilinear')` (align_corners=False), the flipped prediction's left/right classes are "swapped" through
    `tmp = flip_pred_par` -- an ALIAS, not a copy (:932-938), so channels 14/16/18 receive 15/17/19 and 15/17/19 keep
    their own values --, the map is mirrored back (:939) and the two are averaged (:940)."""
    a = F.interpolate(pred_par, size=(size[-2], size[-1]), mode='bilinear')
    b = F.interpolate(flip_pred_par, size=(size[-2], size[-1]), mode='bilinear')
    tmp = b
    for lo, hi in ((14, 15), (16, 17), (18, 19)):
        b[:, lo] = tmp[:, hi]
        b[:, hi] = tmp[:, lo]
    b = b.flip(3)
    return 0.5 * (a + b)


def confusion_matrix(label, pred, num_class, ignore=255):
    """utils/utils.py:190-216 `get_confusion_matrix`: arg-max over classes (first maximum), pixels whose label equals
    `ignore` dropped, counts[label, prediction]."""
    seg_pred = pred.detach().numpy().transpose(0, 2, 3, 1).argmax(axis=3).astype(np.uint8)
    seg_gt = label.detach().numpy().astype(np.int64)
    keep = seg_gt != ignore
    idx = seg_gt[keep] * num_class + seg_pred[keep]
    return np.bincount(idx, minlength=num_class * num_class)[:num_class * num_class].reshape(num_class, num_class).astype(np.float64)
