"""Device-side input hand-off: batched counterparts of the reference's per-sample host functions

    dataset/target_generation.py:94-117   gen_pose_target(joints, visibility, stride, grid_x, grid_y, sigma, aux)
    dataset/target_generation.py:210-239  generate_edge(label, edge_width=3)
    augment_lip_sync.py:127-130           transforms.ToTensor() + Normalize(mean, std)

with the same names and argument meaning, over a whole batch that already sits in HBM: a loader ships uint8 images, uint8
parsing labels and joint coordinates, the three kernels of csrc/targets.hip produce what `train()` consumes
(core/function.py:72-84).  CUDA tensors only -- there is no CPU path here (the CPU restatement is oracle/input_oracle.py,
test infrastructure).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from ._lib import check, desc, lib, new_nhwc, stream_ptr

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # augment_lip_sync.py:127
IMAGENET_STD = (0.229, 0.224, 0.225)


def _need_cuda(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(f"npp_amd.target_generation.{what} runs on the MI355X HIP kernels only (no CPU fallback)")


def gen_pose_target(joints, visibility, stride=8, grid_x=46, grid_y=46, sigma=7, aux=False):
    """joints: [N, J, 2] (x, y) in input pixels, visibility: [N, J] -> (maps [N, J+1, grid_y, grid_x] f32, aux maps | None).
    Channel J is the background map 1 - max_j; `aux=True` adds the 2*sigma maps (target_generation.py:109-117)."""
    _need_cuda(joints, "gen_pose_target")
    j = joints.detach().to(torch.float32).contiguous()
    v = (visibility.detach() != 0).to(torch.uint8).contiguous().to(j.device)
    n, nj = int(j.shape[0]), int(j.shape[1])
    if tuple(j.shape) != (n, nj, 2) or tuple(v.shape) != (n, nj):
        raise ValueError("gen_pose_target: joints must be [N, J, 2] and visibility [N, J]")

    def one(sig):
        maps = torch.empty((n, nj + 1, int(grid_y), int(grid_x)), dtype=torch.float32, device=j.device)
        check(lib().npp_pose_targets(j.data_ptr(), v.data_ptr(), n, nj, int(grid_x), int(grid_y), float(stride), float(sig),
                                     maps.data_ptr(), stream_ptr()), "npp_pose_targets")
        return maps
    return one(sigma), (one(2 * sigma) if aux else None)


def generate_edge(label, edge_width=3, ignore=255, mark_ignore=False):
    """label: uint8 [N, H, W] (or [H, W]) -> uint8 edge map of the same shape (target_generation.py:210-239).
    mark_ignore=True also applies `parsing_edge[parsing_target == 255] = 255` (data_loader.py:284)."""
    _need_cuda(label, "generate_edge")
    squeeze = label.dim() == 2
    lab = (label.unsqueeze(0) if squeeze else label).detach().to(torch.uint8).contiguous()
    n, h, w = (int(s) for s in lab.shape)
    edge = torch.empty_like(lab)
    check(lib().npp_edge_target(lab.data_ptr(), n, h, w, int(edge_width), int(ignore), int(bool(mark_ignore)), edge.data_ptr(),
                                stream_ptr()), "npp_edge_target")
    return edge[0] if squeeze else edge


def normalize_image(images, dtype=torch.float32, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """uint8 RGB [N, H, W, 3] -> logical [N, 3, H, W] in the network's NHWC input layout, (v/255 - mean) / std."""
    _need_cuda(images, "normalize_image")
    if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[3] != 3:
        raise ValueError("normalize_image: expected a uint8 [N, H, W, 3] batch")
    img = images.detach().contiguous()
    n, h, w, _ = (int(s) for s in img.shape)
    out = new_nhwc(n, 3, h, w, dtype, img.device)
    m = (C.c_float * 3)(*[float(a) for a in mean])
    s = (C.c_float * 3)(*[float(a) for a in std])
    check(lib().npp_normalize_image(img.data_ptr(), n, h, w, m, s, C.byref(desc(out)), stream_ptr()), "npp_normalize_image")
    return out
