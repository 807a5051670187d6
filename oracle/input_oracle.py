"""CPU restatement (numpy) of the reference's per-sample target generation -- TEST INFRASTRUCTURE ONLY: nothing under npp_amd/
imports this file; only tests/ may.

PARITY UNPINNED for this row: dataset/target_generation.py imports cv2 at module level and cv2 is absent from this image
(SURVEY §8c lists dataset.* as not importable), so the functions below are restated from the source text and checked against
hand-derived values (tests/test_targets.py), not against outputs of the reference itself.  cv2.dilate with a rectangular kernel
and the default border is restated as a box maximum that ignores out-of-image neighbours.

  gen_single_gaussian_map  dataset/target_generation.py:145-168
  gen_pose_target          dataset/target_generation.py:94-117
  generate_edge            dataset/target_generation.py:210-239 (+ dataset/data_loader.py:281-285 for mark_ignore)
  normalize_image          torchvision ToTensor + Normalize as configured in augment_lip_sync.py:127-130
"""
import numpy as np


def gen_single_gaussian_map(center, stride, grid_x, grid_y, sigma):
    """:145-168 -- the start/end window of the original only skips cells whose exponent exceeds 4.6052 anyway."""
    start = stride / 2.0 - 0.5
    xs = start + np.arange(grid_x, dtype=np.float64) * stride
    ys = start + np.arange(grid_y, dtype=np.float64) * stride
    d2 = (xs[None, :] - float(center[0])) ** 2 + (ys[:, None] - float(center[1])) ** 2
    ex = d2 / 2.0 / sigma / sigma
    g = np.where(ex > 4.6052, 0.0, np.exp(-ex))
    return np.minimum(g, 1.0)


def gen_pose_target(joints, visibility, stride=8, grid_x=46, grid_y=46, sigma=7, aux=False):
    """:94-117 for one sample: joints [J, 2], visibility [J] -> maps [J+1, gy, gx] f64 (+ the 2*sigma maps)."""
    def one(sig):
        jn = joints.shape[0]
        maps = np.zeros((jn + 1, grid_y, grid_x))
        for ji in range(jn):
            if visibility[ji]:
                maps[ji] = gen_single_gaussian_map(joints[ji], stride, grid_x, grid_y, sig)
        maps[jn] = 1 - maps[:jn].max(0)
        return maps
    return one(sigma), (one(2 * sigma) if aux else None)


def generate_edge(label, edge_width=3, ignore=255, mark_ignore=False):
    """:210-239 for one [H, W] label map."""
    label = np.asarray(label)
    h, w = label.shape
    edge = np.zeros((h, w), dtype=np.float64)
    ok = label != ignore
    edge[1:h, :][(label[1:h, :] != label[:h - 1, :]) & ok[1:h, :] & ok[:h - 1, :]] = 1
    edge[:, :w - 1][(label[:, :w - 1] != label[:, 1:w]) & ok[:, :w - 1] & ok[:, 1:w]] = 1
    edge[:h - 1, :w - 1][(label[:h - 1, :w - 1] != label[1:h, 1:w]) & ok[:h - 1, :w - 1] & ok[1:h, 1:w]] = 1
    edge[:h - 1, 1:w][(label[:h - 1, 1:w] != label[1:h, :w - 1]) & ok[:h - 1, 1:w] & ok[1:h, :w - 1]] = 1
    r = edge_width // 2
    pad = np.zeros((h + 2 * r, w + 2 * r))
    pad[r:r + h, r:r + w] = edge
    out = np.zeros((h, w))
    for dy in range(edge_width):
        for dx in range(edge_width):
            out = np.maximum(out, pad[dy:dy + h, dx:dx + w])
    out = out.astype(np.uint8)
    if mark_ignore:
        out[label == ignore] = ignore
    return out


def normalize_image(img_u8, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """uint8 [H, W, 3] -> float32 [3, H, W]"""
    x = img_u8.astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))
