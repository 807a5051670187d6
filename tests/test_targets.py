"""Input hand-off row: the numpy oracle against hand-derived values (CPU), the HIP kernels against the oracle (GPU)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import input_oracle as O


def test_oracle_gaussian_known_values():
    # stride 4, cells at 1.5 + 4g; a joint exactly on cell (gx=3, gy=2) -> 1.0 there, exp(-16/98) one cell away (sigma 7)
    g = O.gen_single_gaussian_map((13.5, 9.5), 4, 12, 10, 7)
    assert g[2, 3] == 1.0
    assert abs(g[2, 4] - np.exp(-16.0 / 98.0)) < 1e-15 and abs(g[3, 3] - np.exp(-16.0 / 98.0)) < 1e-15
    # cut-off: exponent > 4.6052 <=> d2 > 451.3; cell (gx=9) is 24 px away (576 > 451.3) -> 0, cell (gx=8) 20 px (400) -> kept
    assert g[2, 9] == 0.0 and g[2, 8] > 0.0
    maps, aux = O.gen_pose_target(np.array([[13.5, 9.5], [0.0, 0.0]]), np.array([1, 0]), 4, 12, 10, 7, aux=True)
    assert maps.shape == (3, 10, 12) and np.all(maps[1] == 0) and maps[2, 2, 3] == 0.0
    assert np.allclose(maps[2], 1 - maps[0]) and aux[0, 2, 8] > maps[0, 2, 8]


def test_oracle_edge_known_values():
    lab = np.zeros((6, 7), np.uint8)
    lab[:, 4:] = 3                       # vertical boundary between x=3 and x=4
    e1 = O.generate_edge(lab, edge_width=1)
    # raw edge: pixel differs from its right neighbour (x=3), from below-right (x=3, y<5), from below-left (x=4, y<5)
    exp = np.zeros((6, 7), np.uint8)
    exp[:, 3] = 1
    exp[:5, 4] = 1
    assert np.array_equal(e1, exp)
    e3 = O.generate_edge(lab, edge_width=3)
    assert np.array_equal(np.nonzero(e3.any(0))[0], np.array([2, 3, 4, 5]))
    lab[2, 3] = 255                      # ignore pixels take part in no comparison and are marked on request
    e = O.generate_edge(lab, edge_width=1, mark_ignore=True)
    assert e[2, 3] == 255 and e[1, 3] == 1 and e[2, 4] == 1 and e[1, 4] == 0


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X (no GPU visible)")
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_pose_targets_match_oracle():
    from npp_amd import target_generation as TG
    rng = np.random.default_rng(5)
    n, J = 3, 16
    joints = rng.uniform(-20, 400, size=(n, J, 2)).astype(np.float32)
    vis = (rng.uniform(size=(n, J)) > 0.25).astype(np.uint8)
    maps, aux = TG.gen_pose_target(torch.from_numpy(joints).to(_dev()), torch.from_numpy(vis).to(_dev()), stride=4, grid_x=96,
                                   grid_y=96, sigma=7, aux=True)
    assert maps.shape == (n, J + 1, 96, 96) and aux.shape == maps.shape
    for i in range(n):
        r, ra = O.gen_pose_target(joints[i].astype(np.float64), vis[i], 4, 96, 96, 7, aux=True)
        assert np.abs(maps[i].cpu().numpy() - r.astype(np.float32)).max() <= 2e-7      # f64 exp on both sides, one f32 rounding
        assert np.abs(aux[i].cpu().numpy() - ra.astype(np.float32)).max() <= 2e-7


@pytest.mark.gpu
@pytest.mark.parametrize("width", [1, 3, 5])
def test_edge_target_matches_oracle_exactly(width):
    from npp_amd import target_generation as TG
    rng = np.random.default_rng(9)
    n, h, w = 2, 61, 47
    lab = (rng.integers(0, 4, size=(n, h // 6 + 1, w // 5 + 1)).repeat(6, 1).repeat(5, 2)[:, :h, :w]).astype(np.uint8)
    lab[rng.uniform(size=lab.shape) < 0.03] = 255
    lab[:, :3] = 255
    for mark in (False, True):
        e = TG.generate_edge(torch.from_numpy(lab).to(_dev()), edge_width=width, mark_ignore=mark).cpu().numpy()
        for i in range(n):
            assert np.array_equal(e[i], O.generate_edge(lab[i], width, mark_ignore=mark))
    one = TG.generate_edge(torch.from_numpy(lab[0]).to(_dev()), edge_width=width).cpu().numpy()
    assert np.array_equal(one, O.generate_edge(lab[0], width))


@pytest.mark.gpu
def test_normalize_image_matches_oracle():
    from npp_amd import target_generation as TG
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, size=(2, 33, 29, 3), dtype=np.uint8)
    ref = np.stack([O.normalize_image(img[i]) for i in range(2)])
    out = TG.normalize_image(torch.from_numpy(img).to(_dev()), torch.float32)
    assert tuple(out.shape) == (2, 3, 33, 29)
    assert np.abs(out.cpu().numpy() - ref).max() <= 1e-6
    ob = TG.normalize_image(torch.from_numpy(img).to(_dev()), torch.bfloat16).float().cpu().numpy()
    assert np.abs(ob - ref).max() <= 2e-2
    with pytest.raises(RuntimeError):
        TG.normalize_image(torch.from_numpy(img), torch.float32)
