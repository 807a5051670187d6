"""NPPNet search supernet on the HIP kernels -- drop-in for the reference's `models/model_search_interact.py`
(BASELINE config 5: every cross-task / fusion edge is a PC-DARTS `MixedOp` with all 7 `PRIMITIVES_INTER`
candidates live on the first half of the channels).

Same surface: `MixedOp(C, stride, up_scale, extra_conv)`, `channel_shuffle`, `PoseCell` / `ParCell(steps,
multiplier, C_prev_prev, C_prev, C_cur, order)`, `Network(cfg, steps=4, multiplier=4)` reading
`cfg.SEARCH.LAYERS / INIT_CHANNELS`, the 12 architecture tensors (`alphas1..4`, `alphas_pose/par`, `betas*`),
`arch_parameters()`, `loss_entropy()`, `btw()`, `genotype()`, identical state-dict keys.  Reference lines are cited
per class (model_search_interact.py).
"""
from __future__ import annotations

import math

import contextlib
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _ops as K
from . import genotypes as gt
from ._ops import BnSide
from .genotypes import PRIMITIVES_INTER, Genotype_fuse, Genotype_inter
from .model_augment import (Cell, Interpolate, ParCell1, PoseCell1, Upsample, _Head, _Layer, _Stem,  # noqa: F401
                            get_compute_dtype)
from .operations import OPS, _use_batch_stats

BN_MOMENTUM = 0.1


def channel_shuffle(x, groups):
    """model_search_interact.py:22-36.  For the groups == 2 case the supernet uses, see `_ops.interleave2`."""
    n, c, h, w = x.shape
    if groups == 2:
        return K.interleave2(x[:, :c // 2], x[:, c // 2:])
    raise NotImplementedError("channel_shuffle: only groups == 2 is used by the supernet")


class _OpThenBN(nn.Sequential):
    """nn.Sequential(op, nn.BatchNorm2d(C, affine=False)) of MixedOp's pooling candidates (:48-49), run on the HIP
    kernels whatever the BN child has been converted to."""

    def forward(self, x):
        y = self[0](x)
        bn = self[1]
        return K.bn_add(BnSide(y, bn, None), None, relu=False, training=bn.training)


class MixedOp(nn.Module):
    """PC-DARTS mixed edge, model_search_interact.py:39-74: the 7 candidates act on the first C/2 channels, their
    softmax-weighted sum is concatenated with the (resampled) untouched half and channel-shuffled."""

    def __init__(self, C, stride, up_scale=None, extra_conv=None):
        super().__init__()
        self._ops = nn.ModuleList()
        self.mp = nn.MaxPool2d(2, 2)
        for primitive in PRIMITIVES_INTER:
            op = OPS[primitive](C // 2, stride, False)
            if 'pool' in primitive:
                op = _OpThenBN(op, nn.BatchNorm2d(C // 2, affine=False))
            if up_scale:
                op = nn.Sequential(op, Interpolate(scale_factor=up_scale))
            self._ops.append(op)
        self.up_scale = up_scale
        self.extra_conv = extra_conv

    def _sides(self, xtemp):
        """The candidates as deferred-BatchNorm operands (the trailing Interpolate of an up-scaling edge left out)."""
        from .operations import pending_of
        sides = []
        for op in self._ops:
            inner = op[0] if self.up_scale else op
            if isinstance(inner, _OpThenBN):
                sides.append(BnSide(inner[0](xtemp), inner[1], None))
            else:
                sides.append(pending_of(inner, xtemp))
        return sides

    def forward(self, x, weights):
        xtemp, xtemp2 = K.split_half(x)
        if K.MIX_FUSE and self.training:
            # the 7 BatchNorm applies and the weighted sum as ONE launch (K.mix_bn_sum).  On an up-scaling edge every candidate is
            # followed by the same bilinear Interpolate (:46-47 of the reference): resampling is linear with weights that sum to
            # one, so it commutes with the per-channel affine of the BatchNorms and with the weighted sum -- one Interpolate of the
            # mixed low-resolution map instead of seven
            temp1 = K.mix_bn_sum(weights, self._sides(xtemp), self.training)
            if self.up_scale:
                temp1 = self._ops[0][1](temp1)
        else:
            temp1 = K.weighted_sum(weights, [op(xtemp) for op in self._ops])
        if self.up_scale:
            xtemp2 = K.nearest(xtemp2, self.up_scale)       # F.interpolate default mode (:63-64)
        if temp1.shape[2] != xtemp2.shape[2]:
            xtemp2, _ = K.pool2x2(xtemp2, is_avg=False)      # self.mp (:69)
        ans = K.interleave2(temp1, xtemp2)
        if self.extra_conv is not None:
            ec = self.extra_conv
            ans, _ = K.conv2d(ans, ec.weight, ec.bias, 1, 0, 1, relu_in=False)
        return ans


class _MixedCell(nn.Module):
    """PoseCell / ParCell, model_search_interact.py:332-430."""

    def __init__(self, steps, multiplier, C_prev_prev, C_prev, C_cur, order):
        super().__init__()
        from .operations import ReLUConvBN
        if order == 0:
            self.preprocess0 = ReLUConvBN(C_prev_prev, C_cur, 1, 1, 0, affine=True)
            self.preprocess1 = ReLUConvBN(C_prev, C_cur, 1, 1, 0, affine=True)
            self.preprocess2 = ReLUConvBN(C_cur, C_cur, 1, 1, 0, affine=True)
        else:
            self.preprocess0 = ReLUConvBN(3 * C_prev, C_cur, 1, 1, 0, affine=True)
            self.preprocess1 = ReLUConvBN(4 * C_prev, C_cur, 1, 1, 0, affine=True)
            self.preprocess2 = ReLUConvBN(4 * C_prev, C_cur, 1, 1, 0, affine=True)
        self._steps = steps
        self._multiplier = multiplier
        self.order = order
        self._ops = nn.ModuleList()
        for i in range(steps):
            for j in range(3 + i):
                up = {0: 4, 1: 2}.get(j) if order == 0 else None
                self._ops.append(MixedOp(C_cur, 1, up))

    def forward(self, s0, s1, s2, weights, weights2):
        states = [self.preprocess0(s0), self.preprocess1(s1), self.preprocess2(s2)]
        offset = 0
        # (one autograd node for all rows of `weights` / all per-node slices of `weights2` this cell uses: K.split_rows)
        rows = K.split_rows(weights)
        pieces = K.split_slices(weights2, [len(states) + k for k in range(self._steps)])
        for k in range(self._steps):
            outs = [self._ops[offset + j](h, rows[offset + j]) for j, h in enumerate(states)]
            states.append(K.weighted_sum(pieces[k], outs))
            offset += len(outs)
        if self.order == 0:
            states[0] = K.nearest(states[0], 4)
            states[1] = K.nearest(states[1], 2)
        fea1 = K.concat(states[0:3])
        fea2 = K.concat(states[-self._multiplier:])
        return fea1, fea2


class PoseCell(_MixedCell):
    pass


class ParCell(_MixedCell):
    pass


class Network(nn.Module):
    """model_search_interact.py:432-1089."""

    def __init__(self, cfg, steps=4, multiplier=4):
        super().__init__()
        self._num_classes = cfg.DATASET.NUM_CLASSES
        self._num_joints = cfg.DATASET.NUM_JOINTS
        self._layers = cfg.SEARCH.LAYERS
        self._steps = steps
        self._multiplier = multiplier
        self.C = cfg.SEARCH.INIT_CHANNELS
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
        nin = self.num_inchannels

        # encoder-stage mixed edges (:503-523)
        self._ops1, self._ops2 = nn.ModuleList(), nn.ModuleList()
        for i in range(len(nin)):
            for j in range(1 + i):
                up = 1 / 2 ** (i - j)
                ec1 = nn.Conv2d(nin[3 - j], nin[3 - i], 1) if i != j else None
                ec2 = nn.Conv2d(nin[3 - j], nin[3 - i], 1) if i != j else None
                self._ops1.append(MixedOp(nin[3 - j], 1, up, ec1))
                self._ops2.append(MixedOp(nin[3 - j], 1, up, ec2))

        self.upsamples1, self.upsamples2 = nn.ModuleList(), nn.ModuleList()
        for j in range(len(nin) - 1):
            self.upsamples1 += [Upsample(gt.DECODER.upsample1, gt.DECODER.upsample_concat1, nin[j], nin[j + 1])]
        for j in range(len(nin) - 1):
            self.upsamples2 += [Upsample(gt.DECODER.upsample2, gt.DECODER.upsample_concat2, nin[j], nin[j + 1])]

        # decoder-stage mixed edges (:537-559)
        self.up_ops1, self.up_ops2 = nn.ModuleList(), nn.ModuleList()
        resolution = [1, 1 / 2, 1 / 4, 1 / 8, 1 / 4, 1 / 2, 1]
        channels = [int(2 * C / r) for r in resolution]
        for i in range(len(resolution) - 4):
            for j in range(4 + 1 + i):
                up = resolution[4 + i] / resolution[j]
                ec1 = nn.Conv2d(channels[j], channels[4 + i], 1) if 4 + i != j else None
                ec2 = nn.Conv2d(channels[j], channels[4 + i], 1) if 4 + i != j else None
                self.up_ops1.append(MixedOp(channels[j], 1, up, ec1))
                self.up_ops2.append(MixedOp(channels[j], 1, up, ec2))

        Cf = nin[3]

        def layer(cout):
            return _Layer(nn.ReLU(), nn.Conv2d(8 * Cf, cout, kernel_size=1, padding=0, dilation=1), nn.BatchNorm2d(cout))

        self.pose_layer = layer(4 * Cf)
        self.pose_auxlayer = layer(3 * Cf)
        self.par_layer = layer(4 * Cf)
        self.edge_layer = layer(3 * Cf)

        self.pose_net, self.par_net = nn.ModuleList(), nn.ModuleList()
        for _ in range(3):
            self.pose_net.append(PoseCell(4, 4, Cf, Cf, Cf, 1))
            self.par_net.append(ParCell(4, 4, Cf, Cf, Cf, 1))

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
        self._packer = None
        self._auto = None
        self.init_weights()
        self._initialize_alphas()

    # -- architecture parameters (:772-804) -------------------------------------------------------------------
    def _initialize_alphas(self):
        k = sum(1 for i in range(self._steps) for _ in range(3 + i))
        num_ops = len(PRIMITIVES_INTER)
        self.alphas1 = nn.Parameter(1e-3 * torch.ones(10, num_ops))
        self.alphas2 = nn.Parameter(1e-3 * torch.ones(10, num_ops))
        self.alphas3 = nn.Parameter(1e-3 * torch.ones(18, num_ops))
        self.alphas4 = nn.Parameter(1e-3 * torch.ones(18, num_ops))
        self.betas1 = nn.Parameter(1e-3 * torch.ones(10))
        self.betas2 = nn.Parameter(1e-3 * torch.ones(10))
        self.betas3 = nn.Parameter(1e-3 * torch.ones(18))
        self.betas4 = nn.Parameter(1e-3 * torch.ones(18))
        self.alphas_pose = nn.Parameter(1e-3 * torch.ones(k, num_ops))
        self.alphas_par = nn.Parameter(1e-3 * torch.ones(k, num_ops))
        self.betas_pose = nn.Parameter(1e-3 * torch.ones(k))
        self.betas_par = nn.Parameter(1e-3 * torch.ones(k))
        self._arch_parameters = [self.alphas1, self.alphas2, self.alphas3, self.alphas4, self.alphas_pose, self.alphas_par,
                                 self.betas1, self.betas2, self.betas3, self.betas4, self.betas_pose, self.betas_par]

    def arch_parameters(self):
        return self._arch_parameters

    def btw(self, n_input, steps, betas):
        """beta -> per-node softmax edge weights (:1054-1065)."""
        pieces = K.split_slices(betas, [n_input + k for k in range(steps)])
        return torch.cat([F.softmax(pc, dim=-1) for pc in pieces], dim=0)

    def loss_entropy(self):
        """:881-890 (normalised entropy of the alpha rows)."""
        length = len(self._arch_parameters)
        alphas = self._arch_parameters[0:length // 2]
        en_alphas = 0.
        for a in alphas:
            w1 = F.softmax(a, dim=-1)
            en = -(w1 * torch.log(w1)).sum(-1) / math.log(w1.shape[1])
            en_alphas = en_alphas + en.mean(dim=0)
        return 0.25 * 2 * en_alphas / length

    def entropy_beta(self, n_input, steps, betas):
        """:898-908: mean over the `steps` nodes of the normalised entropy of the softmax over each node's beta slice (the
        slices grow by one entry per node, as in `btw`)."""
        start, n, en = 0, n_input, 0.
        for _ in range(steps):
            w = F.softmax(betas[start:start + n], dim=-1)
            en = en + (-(w * torch.log(w)).sum(-1)) / math.log(n)
            start += n
            n += 1
        return en / steps

    # -- forward (:626-770) -------------------------------------------------------------------------------------
    def _mix(self, ops, base, feats, alpha_rows, beta_rows):
        w = K.split_rows(F.softmax(alpha_rows, dim=-1))      # (one autograd node for the rows: K.split_rows)
        w2 = F.softmax(beta_rows, dim=-1)
        outs = [ops[base + j](h, w[j]) for j, h in enumerate(feats)]
        return K.weighted_sum(w2, outs)

    def forward(self, x):
        """model_search_interact.py:626-770.  NPP_AUTO_GRAPH=1: replayed as hipGraphs after the first calls (npp_amd/auto_graph.py)."""
        from . import auto_graph
        if auto_graph.ENABLED and self.training:
            if self._auto is None:
                self._auto = auto_graph.AutoGraph(self)
            return self._auto(x)
        return self._forward_eager(x)

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_packer"] = None
        d["_auto"] = None
        return d

    def _forward_eager(self, x):
        if not x.is_cuda:
            raise RuntimeError("npp_amd supernet runs on the MI355X HIP kernels only (no CPU fallback)")
        dt = get_compute_dtype()
        K.fan_reset()
        if self._packer is None:
            from .operations import SE_Block
            skip = set()
            for m in self.modules():
                if isinstance(m, SE_Block):
                    skip.update((id(m.conv1.weight), id(m.conv2.weight)))
            # (the merged edges of the fixed encoder cells get their images from the packer too, as in model_augment.Network: without
            # the groups every merged conv rebuilt its images with a cat + a zero fill + a copy per member, ~120 launches per step)
            groups = [g for m in self.modules() for g in getattr(m, "_wide_groups", ())] if K.WIDE else []
            self._packer = K.WeightPacker((m.weight for m in self.modules()
                                           if isinstance(m, nn.Conv2d) and m.groups == 1 and id(m.weight) not in skip), groups)
        self._packer.pack_if_stale(dt, x.device, force=self.training)
        if self.training:
            K.note_training_step()
        x = K.image_to_nhwc(x, dt)
        # two task branches on two HIP streams, as in model_augment.Network.forward (the pose branch on the caller's)
        from .model_augment import _side_stream, _stream_mode, Network as _AugNet
        # SyncBatchNorm (search_lip_sync.py:268-271): with the statistics going through the peer-to-peer mailboxes (csrc/p2p.hip) an
        # exchange is an ordinary kernel on the stream that needs it and the two branch streams stay; with collectives (which must
        # all sit on one stream) the supernet runs on one stream
        sync_bn = _AugNet._sync_bn_active(self)
        K.P2P_DIRECT = bool(sync_bn and x.is_cuda and os.environ.get("NPP_SYNCBN_STREAMS") is None
                            and os.environ.get("NPP_P2P_DIRECT", "1") != "0" and _AugNet._p2p_ready(self))
        two = _stream_mode() >= 2 and (not sync_bn or K.P2P_DIRECT)
        K._hub_offload = None
        K._hub_stream = None
        if two:
            sa = torch.cuda.current_stream()
            sb = _side_stream(x.device, 0)
            K._hub_stream = sa

        def on_b():
            return torch.cuda.stream(sb) if two else contextlib.nullcontext()

        def meet(*tensors):
            if not two:
                return
            ea, eb = torch.cuda.Event(), torch.cuda.Event()
            ea.record(sa)
            eb.record(sb)
            sa.wait_event(eb)
            sb.wait_event(ea)
            for t in tensors:
                t.record_stream(sa)
                t.record_stream(sb)

        meet(x)
        s1 = self.stem2(s0 := self.stem1(self.stem0(x)))
        with on_b():
            s3 = self.stem5(s2 := self.stem4(self.stem3(x)))
        f1, f2 = [], []
        offset = 0
        # the row blocks of the interaction parameters every tap / decoder stage uses, through ONE autograd node per tensor
        # (K.split_slices; the reference slices `alphas1[offset:offset + n]` stage by stage, :655-700)
        ntap = len(self._taps)
        sz1, sz3 = list(range(1, ntap + 1)), [ntap + 1 + d for d in range(3)]
        A1, B1, A2, B2 = (K.split_slices(t, sz1) for t in (self.alphas1, self.betas1, self.alphas2, self.betas2))
        A3, B3, A4, B4 = (K.split_slices(t, sz3) for t in (self.alphas3, self.betas3, self.alphas4, self.betas4))
        for i, (cell1, cell2) in enumerate(zip(self.cells1, self.cells2)):
            s0, s1 = s1, cell1(s0, s1)
            with on_b():
                s2, s3 = s3, cell2(s2, s3)
            if i in self._taps:
                f1.append(s1)
                f2.append(s3)
                n = len(f1)
                meet(*f1, *f2)
                s1 = K.add(s1, self._mix(self._ops1, offset, f2, A1[n - 1], B1[n - 1]))
                with on_b():
                    s3 = K.add(s3, self._mix(self._ops2, offset, f1, A2[n - 1], B2[n - 1]))
                f1[-1], f2[-1] = s1, s3
                offset += n
        cont = 0
        for d in range(3):
            o1 = self.upsamples1[d](f1[3] if d == 0 else f1[-1], f1[2 - d])
            with on_b():
                o2 = self.upsamples2[d](f2[3] if d == 0 else f2[-1], f2[2 - d])
            f1.append(o1)
            f2.append(o2)
            n = len(f1)
            meet(*f1, *f2)
            g1 = list(f1)        # both mixes read the features as they are BEFORE this stage's adds
            f1[-1] = K.add(o1, self._mix(self.up_ops1, cont, f2, A3[d], B3[d]))
            with on_b():
                f2[-1] = K.add(o2, self._mix(self.up_ops2, cont, g1, A4[d], B4[d]))
            cont += n
        H, W = f1[0].shape[2], f1[0].shape[3]
        x1 = K.concat([f1[0], f1[6], K.bilinear(f1[5], H, W), K.bilinear(f1[4], H, W)])
        in1, in3 = self.pose_auxlayer(x1), self.pose_layer(x1)
        with on_b():
            x2 = K.concat([f2[0], f2[6], K.bilinear(f2[5], H, W), K.bilinear(f2[4], H, W)])
            in2, in4 = self.edge_layer(x2), self.par_layer(x2)
        pose_list, par_list = [], []

        def heads(i):
            pose_aux = self.pose_auxnet[i](in1)
            pose_map = self.pose_head[i](in3)
            with on_b():
                edge = self.edge_head[i](in2)
                par_map = self.par_head[i](in4)
            pose_list.append([pose_map, pose_aux])
            par_list.append([par_map, edge])

        heads(0)
        w_pose = F.softmax(self.alphas_pose, dim=-1)
        w_pose2 = self.btw(3, self._steps, self.betas_pose)
        w_par = F.softmax(self.alphas_par, dim=-1)
        w_par2 = self.btw(3, self._steps, self.betas_par)
        for i in range(1, self.refine_layers + 1):
            for j in range(3):
                m = 2 * (i - 1) + j
                meet(in1, in2, in3, in4, w_par, w_par2)
                n1, tmp = self.pose_net[m](in1, in3, in4, w_pose, w_pose2)
                with on_b():
                    in2, n4 = self.par_net[m](in2, in3, in4, w_par, w_par2)
                in1, in3, in4 = n1, tmp, n4
            heads(i)
        if two:
            sa.wait_stream(sb)
            for pair in par_list:
                for t in pair:
                    t.record_stream(sa)
        K.fan_reset()
        return pose_list, par_list

    # -- genotype extraction (:913-1052; host-side numpy on the architecture tensors) --------------------------------
    def genotype(self):
        def parse_inter(w1, w2, n_input, step):
            gene, start, n = [], 0, n_input
            for _ in range(step):
                Wm = w1[start:start + n].copy() * w2[start:start + n].copy()[:, None]
                prob, picks = 0., []
                while prob < 0.7 and len(picks) < 4:
                    m = np.max(Wm)
                    prob += m
                    idx = np.where(Wm == m)
                    Wm[idx] = 0
                    picks.append((PRIMITIVES_INTER[idx[1][0]], int(idx[0][0])))
                gene.append(picks)
                start += n
                n += 1
            return gene

        def parse_fuse(w1, w2):
            gene, start, n = [], 0, 3
            for i in range(self._steps):
                Wm = w1[start:start + n].copy() * w2[start:start + n].copy()[:, None]
                edges = sorted(range(i + 3), key=lambda e: -max(Wm[e]))[:2]
                for j in edges:
                    gene.append((PRIMITIVES_INTER[int(np.argmax(Wm[j]))], j))
                start += n
                n += 1
            return gene

        def sm(a):
            return F.softmax(a, dim=-1).data.cpu().numpy()

        inter = Genotype_inter(
            task1=parse_inter(sm(self.alphas1), self.btw(1, 4, self.betas1).data.cpu().numpy(), 1, 4),
            task2=parse_inter(sm(self.alphas2), self.btw(1, 4, self.betas2).data.cpu().numpy(), 1, 4),
            task3=parse_inter(sm(self.alphas3), self.btw(5, 3, self.betas3).data.cpu().numpy(), 5, 3),
            task4=parse_inter(sm(self.alphas4), self.btw(5, 3, self.betas4).data.cpu().numpy(), 5, 3))
        fuse = Genotype_fuse(
            pose=parse_fuse(sm(self.alphas_pose), self.btw(3, self._steps, self.betas_pose).data.cpu().numpy()),
            pose_concat=range(3, 7),
            par=parse_fuse(sm(self.alphas_par), self.btw(3, self._steps, self.betas_par).data.cpu().numpy()),
            par_concat=range(3, 7))
        return inter, fuse

    def init_weights(self, pretrained=''):
        """:1067-1089"""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight.data)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.modules.batchnorm._BatchNorm):
                if m.affine:
                    m.weight.data.fill_(1)
                    m.bias.data.zero_()
