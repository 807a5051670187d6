"""Device-side evaluation helpers: the parsing half of `validate_sync` (core/function.py:906-967).

    cm = ParsingConfusion(num_classes=20, ignore=255)
    for images, label_par, ... in loader:
        pred_pose, pred_par = model(images);  flip_pose, flip_par = model(images.flip(3))
        cm.update(pred_par[-1][0], flip_par[-1][0], label_par[0])        # no logits leave the GPU
    confusion = cm.matrix()             # == sum of utils.utils.get_confusion_matrix(...) over the batches

`alias_swap=True` (default) reproduces the reference bit for bit, including its aliased left/right channel swap
(function.py:932-938 assign through `tmp = flip_pred_par`, so channels 15/17/19 are never replaced); pass False for the
swap the authors presumably meant.
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import _ops as K
from ._lib import check, lib, stream_ptr


class ParsingConfusion:
    def __init__(self, num_classes=20, ignore=255, alias_swap=True):
        self.c, self.ignore, self.alias = int(num_classes), int(ignore), bool(alias_swap)
        self.counts = None

    def update(self, pred, flip_pred, label):
        if not pred.is_cuda:
            raise RuntimeError("npp_amd.evaluate runs on the MI355X HIP kernels only (no CPU fallback)")
        if pred.shape[1] != self.c:
            raise ValueError(f"expected {self.c} classes, got {pred.shape[1]}")
        pred = L.to_nhwc(pred.detach())
        fl = L.to_nhwc(flip_pred.detach()) if flip_pred is not None else None
        if fl is not None and fl.dtype != pred.dtype:
            fl = K.cast(fl, pred.dtype)
        label = label.to(device=pred.device, dtype=torch.int64).contiguous()
        n, H, W = label.shape
        if n != pred.shape[0]:
            raise ValueError("batch mismatch between logits and labels")
        if self.counts is None:
            self.counts = torch.zeros(self.c * self.c, dtype=torch.int64, device=pred.device)
        check(lib().npp_parsing_confusion(K._byref(pred), K._byref(fl) if fl is not None else None, label.data_ptr(), H, W,
                                          self.ignore, int(self.alias), self.counts.data_ptr(), stream_ptr()),
              "npp_parsing_confusion")

    def matrix(self):
        if self.counts is None:
            return torch.zeros(self.c, self.c, dtype=torch.float64)
        return self.counts.view(self.c, self.c).to(torch.float64).cpu()
