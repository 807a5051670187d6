"""Loss heads on the HIP kernels -- drop-in for the reference's `core/criterion.py`.

Same classes / constructor arguments / `lamda` parameters: `Criterion_pose(out_len, use_target_weight)`
(core/criterion.py:74-145), `OhemCrossEntropy` (:43-72), `Criterion_par(out_len, ignore_index, thres,
min_kept)` (:148-217).  The per-pixel work (bilinear x4 upsample of the logits to the label size, softmax,
weighted NLL, OHEM selection, and all of their backward) runs in `csrc/loss.hip`; the upsampled
[N,C,384,384] logits are never materialised and OHEM's global sort is an exact radix select.  Only the
scalar tail (`* exp(-lamda) + lamda`, the final mean) is torch arithmetic on 0-d tensors.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _ops as K

# LIP class weights, core/criterion.py:17-21
lip = [0.7602572, 0.94236198, 0.85644457, 1.04346266, 1.10627293, 0.80980162,
       0.95168713, 0.8403769, 1.05798412, 0.85746254, 1.01274366, 1.05854692,
       1.03430773, 0.84867818, 0.88027721, 0.87580925, 0.98747462, 0.9876475,
       1.00016535, 1.00108882]
weights_lip = torch.tensor(lip, dtype=torch.float32)
pascal = [0.82877791, 0.95688253, 0.94921949, 1.00538108, 1.0201687, 1.01665831, 1.05470914]
weights_pascal = torch.tensor(pascal, dtype=torch.float32)


class OhemCrossEntropy(nn.Module):
    """core/criterion.py:43-72.  `score` may be at the label resolution or lower (it is then upsampled
    bilinearly with align_corners=True inside the kernel, as Criterion_par does before calling the
    reference's OhemCrossEntropy)."""

    def __init__(self, ignore_index=255, thres=0.7, min_kept=100000, weight=weights_lip):
        super().__init__()
        self.thresh = thres
        self.min_kept = max(1, min_kept)
        self.ignore_index = ignore_index
        self.register_buffer("class_weight", weight.clone().float(), persistent=False)

    def forward(self, score, target, **kwargs):
        cw = self.class_weight
        if cw.device != score.device:
            cw = cw.to(score.device)
            self.class_weight = cw
        return K.upsampled_ce(score, target, cw, self.ignore_index, ohem=(self.thresh, self.min_kept))


class Criterion_pose(nn.Module):
    """core/criterion.py:74-145."""

    def __init__(self, out_len=1, use_target_weight=False):
        super().__init__()
        self.use_target_weight = use_target_weight
        self.lamda = nn.Parameter(-2.5 * torch.ones(out_len))

    def joint_loss(self, output, target, target_weight=None):
        """sum_j MSE(pred_j, gt_j) over main (+ aux) maps / num_joints, criterion.py:82-128.
        Every per-joint MSE is a mean over N*H*W elements, so the sum over joints is SSE / (N*H*W); with
        `use_target_weight` prediction and target of (image n, joint j) are both scaled by target_weight[n, j]
        (criterion.py:103-108).  A heat-map whose size differs from its target's is resampled like
        `F.interpolate(size=(h, w), mode='bilinear')` (align_corners=False), (h, w) = the MAIN target's size -- also for
        the auxiliary map, exactly as criterion.py:92-96, 113-115 do."""
        if isinstance(output, list):
            outs, tgts = [output[0], output[1]], [target[0], target[1]]
        else:
            outs, tgts = [output], [target[0] if isinstance(target, list) else target]
        J = outs[0].size(1)
        h, w = tgts[0].shape[2:]
        wt = None
        if self.use_target_weight:
            if target_weight is None:
                raise ValueError("Criterion_pose(use_target_weight=True) needs target_weight [N, num_joints, 1]")
            wt = target_weight
        loss = 0.
        for o, t in zip(outs, tgts):
            if tuple(o.shape[2:]) != tuple(t.shape[2:]):
                o = K.bilinear(o, int(h), int(w), align_corners=False)
            loss = loss + K.mse_sse(o, t, wt) / float(o.size(0) * o.size(2) * o.size(3))
        return loss / J

    def forward(self, output, target, target_weight=None):
        loss = 0.
        if isinstance(output, list) and K.FUSED_CRITERIA and isinstance(target, list):
            fused = self._fused(output, target, target_weight)
            if fused is not None:
                return fused
        if isinstance(output, list):
            for i in range(len(output)):
                loss = loss + self.joint_loss(output[i], target, target_weight) * torch.exp(-self.lamda[i]) + self.lamda[i]
        else:
            loss = loss + self.joint_loss(output, target, target_weight) * torch.exp(-self.lamda) + self.lamda
        return loss


    def _fused(self, output, target, target_weight):
        """The whole criterion as one autograd node (K.criterion_fused): every MSE term + the scalar tail, same arithmetic."""
        specs, xs = [], []
        wt = None
        if self.use_target_weight:
            if target_weight is None:
                raise ValueError("Criterion_pose(use_target_weight=True) needs target_weight [N, num_joints, 1]")
            wt = target_weight
        for i, stage_out in enumerate(output):
            outs = [stage_out[0], stage_out[1]] if isinstance(stage_out, list) else [stage_out]
            tgts = [target[0], target[1]] if isinstance(stage_out, list) else [target[0] if isinstance(target, list) else target]
            J = outs[0].size(1)
            h, w = tgts[0].shape[2:]
            for o, t in zip(outs, tgts):
                if tuple(o.shape[2:]) != tuple(t.shape[2:]):
                    o = K.bilinear(o, int(h), int(w), align_corners=False)
                specs.append(("mse", t, wt, 1.0 / (float(o.size(0) * o.size(2) * o.size(3)) * J), i))
                xs.append(o)
        if len(xs) > 32:
            return None
        return K.criterion_fused(self.lamda, specs, xs)


class Criterion_par(nn.Module):
    """core/criterion.py:148-217."""

    def __init__(self, out_len=1, ignore_index=255, thres=0.9, min_kept=131072):
        super().__init__()
        self.ignore_index = ignore_index
        self.criterion = OhemCrossEntropy(ignore_index=ignore_index, thres=thres, min_kept=min_kept, weight=weights_lip)
        self.lamda = nn.Parameter(2.3 * torch.ones(out_len))

    def parsing_loss(self, preds, target, edge_w=None):
        """criterion.py:158-202: OHEM CE on the parsing logits + class-balanced CE on the edge logits, both on
        logits upsampled to the label size."""
        if edge_w is None:
            edge_w = K.edge_class_weights(target[1])
        loss = 0.
        if isinstance(preds, list):
            par = preds[0]
            if isinstance(par, list):
                loss = loss + self.criterion(par[0], target[0]) + self.criterion(par[1], target[0]) * 0.4
            else:
                loss = loss + self.criterion(par, target[0])
            edges = preds[1] if isinstance(preds[1], list) else [preds[1]]
            for e in edges:
                loss = loss + K.upsampled_ce(e, target[1], edge_w, self.ignore_index, ohem=None)
        else:
            loss = loss + self.criterion(preds, target[0])
        return loss

    def _fused(self, preds, target):
        """The whole criterion as one autograd node (K.criterion_fused)."""
        crit = self.criterion
        cw = crit.class_weight
        if cw.device != target[0].device:
            cw = cw.to(target[0].device)
            crit.class_weight = cw
        edge_w = K.edge_class_weights_dev(target[1])
        specs, xs = [], []
        ohem = (crit.thresh, crit.min_kept)
        for i, p in enumerate(preds):
            if isinstance(p, list):
                par = p[0]
                if isinstance(par, list):
                    specs += [("ce", target[0], cw, crit.ignore_index, ohem, 1.0, i), ("ce", target[0], cw, crit.ignore_index, ohem, 0.4, i)]
                    xs += [par[0], par[1]]
                else:
                    specs.append(("ce", target[0], cw, crit.ignore_index, ohem, 1.0, i))
                    xs.append(par)
                for e in (p[1] if isinstance(p[1], list) else [p[1]]):
                    specs.append(("ce", target[1], edge_w, self.ignore_index, None, 1.0, i))
                    xs.append(e)
            else:
                specs.append(("ce", target[0], cw, crit.ignore_index, ohem, 1.0, i))
                xs.append(p)
        if len(xs) > 32:
            return None
        return K.criterion_fused(self.lamda, specs, xs)

    def forward(self, preds, target):
        loss = 0.
        if isinstance(preds, list) and K.FUSED_CRITERIA and target[0].is_cuda:
            fused = self._fused(preds, target)
            if fused is not None:
                return fused
        if isinstance(preds, list):
            edge_w = K.edge_class_weights(target[1])
            for i in range(len(preds)):
                loss = loss + self.parsing_loss(preds[i], target, edge_w) * torch.exp(-self.lamda[i]) + self.lamda[i]
        else:
            loss = loss + self.criterion(preds, target) * torch.exp(-self.lamda) + self.lamda
        return loss
