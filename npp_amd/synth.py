"""Deterministic synthetic weights and LIP-shaped batches.

The reference ships no images and no checkpoints (`/root/reference/.MISSING_LARGE_BLOBS`),
so every parity fixture, smoke run and bench line in this repo is driven by the
generators below.  They are keyed by *name* (state-dict key / batch field), not by
call order, so the oracle, the golden-vector script and the HIP path all see the
same numbers on any machine without shipping a 294 MiB weight file.

Batch layout follows what `dataset/data_loader.py:281-304` hands to
`core/function.py:72-84`.
"""
from __future__ import annotations

import zlib
import numpy as np

SEED_BASE = 0x4E5050  # "NPP"


def _rng(key: str, seed: int = 0) -> np.random.Generator:
    h = zlib.crc32(key.encode("utf-8")) & 0xFFFFFFFF
    return np.random.Generator(np.random.PCG64([SEED_BASE, seed & 0xFFFFFFFF, h]))


def synth_state_dict(template: dict, seed: int = 0) -> dict:
    """Fill every entry of a state-dict-shaped {key: array-like with .shape} mapping.

    conv weight (4-D)    ~ N(0, 2/(fan_in+fan_out))      (xavier-normal, cf. model_augment.py:655)
    conv bias            ~ N(0, 0.02)
    BN weight            ~ U(0.5, 1.5)     BN bias ~ N(0, 0.1)
    BN running_mean      ~ N(0, 0.1)       running_var ~ U(0.5, 1.5)
    num_batches_tracked  = 0
    Returns {key: np.ndarray} (float32, int64 for the counters).
    """
    keys = set(template.keys())
    out = {}
    for k, t in template.items():
        shape = tuple(t.shape)
        r = _rng(k, seed)
        base, leaf = (k.rsplit(".", 1) + [""])[:2] if "." in k else ("", k)
        is_bn = (base + ".running_mean") in keys
        if leaf == "num_batches_tracked":
            out[k] = np.zeros(shape, dtype=np.int64)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            fan_out = shape[0] * shape[2] * shape[3]
            std = np.sqrt(2.0 / (fan_in + fan_out))
            out[k] = (r.standard_normal(shape) * std).astype(np.float32)
        elif is_bn and leaf == "weight":
            out[k] = r.uniform(0.5, 1.5, shape).astype(np.float32)
        elif is_bn and leaf == "bias":
            out[k] = (r.standard_normal(shape) * 0.1).astype(np.float32)
        elif leaf == "running_mean":
            out[k] = (r.standard_normal(shape) * 0.1).astype(np.float32)
        elif leaf == "running_var":
            out[k] = r.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf == "bias":
            out[k] = (r.standard_normal(shape) * 0.02).astype(np.float32)
        else:  # anything else (lamda, alphas): small normal
            out[k] = (r.standard_normal(shape) * 0.1).astype(np.float32)
    return out


def _gauss_maps(r, n, joints, hm, sigma):
    """`joints`+1 heat-maps per image on an hm x hm grid: Gaussians at seeded joint
    positions, clipped below e^-4.6052 (cf. dataset/target_generation.py:145-168),
    last channel = background = 1 - max over joints."""
    ys, xs = np.meshgrid(np.arange(hm, dtype=np.float32), np.arange(hm, dtype=np.float32), indexing="ij")
    out = np.zeros((n, joints + 1, hm, hm), dtype=np.float32)
    cx = r.uniform(0.1 * hm, 0.9 * hm, (n, joints)).astype(np.float32)
    cy = r.uniform(0.1 * hm, 0.9 * hm, (n, joints)).astype(np.float32)
    for b in range(n):
        for j in range(joints):
            d2 = (xs - cx[b, j]) ** 2 + (ys - cy[b, j]) ** 2
            e = d2 / (2.0 * sigma * sigma)
            g = np.exp(-e)
            g[e > 4.6052] = 0.0
            out[b, j] = g
        out[b, joints] = 1.0 - out[b, :joints].max(axis=0)
    return out, (cx, cy)


def synth_batch(n: int, size: int = 384, num_classes: int = 20, num_joints: int = 16,
                seed: int = 0, rank: int = 0):
    """One training batch in the reference's tuple layout.

    images      [n,3,size,size] float32, N(0,1) (ImageNet-normalised pixels are ~N(0,1))
    labels_par  [par [n,size,size] int64 in {0..C-1, 255}, edge [n,size,size] int64 in {0,1,255}]
                rows 0..7 carry the ignore label 255
    labels_pose [hm [n,J+1,size/4,size/4] float32 (sigma 7/4 px on the strided grid... scaled), hm_aux (2x sigma)]
                the last channel is background; the train step drops it (core/function.py:81-82)
    meta        {'pose_weight': [n,J,1] float32 ones}
    All numpy; callers move to torch/device.
    """
    r = _rng("batch", seed * 1000003 + rank)
    hm = size // 4
    images = r.standard_normal((n, 3, size, size)).astype(np.float32)
    # parsing labels: piecewise-constant blocks (so bilinear-upsampled logits face real regions)
    blk = max(size // 16, 1)
    coarse = r.integers(0, num_classes, (n, (size + blk - 1) // blk, (size + blk - 1) // blk))
    par = np.repeat(np.repeat(coarse, blk, axis=1), blk, axis=2)[:, :size, :size].astype(np.int64)
    par[:, :8, :] = 255
    edge = (r.random((n, size, size)) < 0.1).astype(np.int64)
    edge[:, :8, :] = 255
    sig = 7.0 * hm / 96.0
    hm_main, _ = _gauss_maps(r, n, num_joints, hm, sig)
    hm_aux, _ = _gauss_maps(r, n, num_joints, hm, 2.0 * sig)
    meta = {"pose_weight": np.ones((n, num_joints, 1), dtype=np.float32)}
    return images, [par, edge], [hm_main, hm_aux], meta


def synth_batch_hw(n: int, h: int, w: int, seed: int = 0, rank: int = 0):
    """A non-square batch: the top-left h x w window of the square batch of side max(h, w) (heat-maps: h/4 x w/4)."""
    size = max(h, w)
    images, (par, edge), (hm_main, hm_aux), meta = synth_batch(n, size, seed=seed, rank=rank)
    return (np.ascontiguousarray(images[:, :, :h, :w]),
            [np.ascontiguousarray(par[:, :h, :w]), np.ascontiguousarray(edge[:, :h, :w])],
            [np.ascontiguousarray(hm_main[:, :, :h // 4, :w // 4]), np.ascontiguousarray(hm_aux[:, :, :h // 4, :w // 4])], meta)
