"""Shared helpers for the parity tests."""
import os
import numpy as np
import torch

from npp_amd.synth import synth_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


class _Shape:
    def __init__(self, shape):
        self.shape = shape


def template_from_golden(g):
    keys = [str(k) for k in g["sd_keys"]]
    shapes = [tuple(int(d) for d in str(s).split(",") if d != "") for s in g["sd_shapes"]]
    return {k: _Shape(s) for k, s in zip(keys, shapes)}


def synth_tensors(template, seed=0, dtype=torch.float32, prefix=""):
    tmpl = {prefix + k: v for k, v in template.items()}
    syn = synth_state_dict(tmpl, seed)
    out = {}
    for k in template:
        a = syn[prefix + k]
        t = torch.from_numpy(a)
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out


def rel_err(a, b):
    """max |a-b| / max|b| -- the 'relative fp32' measure used for the 1e-3 parity bar."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def rel_l2(a, b):
    """||a-b|| / ||b||: the bf16 (storage precision) checks use the norm, because rounding moves arg-max
    ties of max-pool and individual elements by O(1) while the tensor as a whole stays close."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
