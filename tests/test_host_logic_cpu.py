"""Host-side logic added in round 5 that needs no GPU: the background runner of the GPU suite's child processes, the bookkeeping of the
SyncBatchNorm exchanges that are folded into the fused BatchNorm kernels (which sides leave the sync pool, what a later flush sends),
and the launch heuristics that are plain arithmetic."""
import os
import sys
import types

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def test_background_children_run_two_at_a_time_and_keep_their_own_ports(tmp_path, monkeypatch):
    import importlib.util
    spec = importlib.util.spec_from_file_location("bg_children_private", os.path.join(REPO, "tests", "bg_children.py"))
    bg = importlib.util.module_from_spec(spec)      # a private registry for this test (the suite's own stays as it is)
    spec.loader.exec_module(bg)
    if True:
        code = "import os, time; time.sleep(0.2); print(os.environ['MASTER_PORT'], os.environ.get('X', '-'))"
        order = []
        bg.before_start(lambda: order.append("before"))
        for k in range(4):
            bg.register(f"child-{k}", [sys.executable, "-c", code], {"X": str(k)} if k % 2 else None, timeout=60)
        assert bg.register("child-0", ["never", "run"]) == "child-0"      # a key is registered once
        r3 = bg.result("child-3")                                           # the first request starts all of them
        assert order == ["before"]
        outs = [bg.result(f"child-{k}") for k in range(4)]
        assert all(r.returncode == 0 for r in outs) and r3 is outs[3]
        ports = [r.stdout.split()[0] for r in outs]
        assert len(set(ports)) == 4, ports                                   # every child its own rendezvous port
        assert [r.stdout.split()[1] for r in outs] == ["-", "1", "-", "3"]


def _side(K, c, stats_c=0, rider=False, synced=0):
    sd = K.BnSide.__new__(K.BnSide)
    for name in K.BnSide.__slots__:
        setattr(sd, name, None)
    sd.x = torch.zeros(1, c, 2, 2)
    sd.bn = torch.nn.BatchNorm2d(c)
    sd.stats = torch.zeros(K.R * 2 * (stats_c or c), dtype=torch.float64)
    sd.stats_c, sd.rider, sd.synced_ws, sd.private = stats_c, rider, synced, True
    return sd


def test_folded_exchange_bookkeeping_without_a_transport_changes_nothing():
    """No mailboxes (CPU, or NPP_SYNCBN_P2P=0): _fold_forward refuses, nothing is marked exchanged, nothing leaves a pool."""
    from npp_amd import _ops as K
    a, b = _side(K, 8), _side(K, 8)
    pool = K._sync_pool.cur()
    pool.waiting = [a, b]
    try:
        assert K._fold_forward([(a, b)], True) == -1
        assert K._fold_channel(32, None) == -1 and not K.fold_ready(None)
        assert a.synced_ws == 0 and b.synced_ws == 0 and pool.waiting == [a, b]
    finally:
        pool.waiting = []


def test_a_side_that_leaves_the_pool_takes_only_itself():
    from npp_amd import _ops as K
    lead, r1, r2 = _side(K, 8, stats_c=24), _side(K, 8, stats_c=24, rider=True), _side(K, 8, stats_c=24, rider=True)
    other = _side(K, 16)
    pool = K._sync_pool.cur()
    pool.waiting = [lead, r1, other, r2]
    try:
        assert K._leave_pool(r1) is pool and pool.waiting == [lead, other, r2]
        assert K._leave_pool(lead) is pool and pool.waiting == [other, r2]
        assert K._leave_pool(lead) is None                                   # not waiting any more: nothing happens
        assert r2.rider and r2.carry is None                                 # (riders are not promoted: a flush sends private copies)
    finally:
        pool.waiting = []


def test_compact_copy_of_a_merged_edge_is_private_and_complete():
    """What a flush sends for an edge of a merged conv once exchanges are folded: its own [R][2C] slice of the [R][2 * stats_c] rows."""
    from npp_amd import _ops as K
    c, sc = 4, 12
    rows = torch.arange(K.R * 2 * sc, dtype=torch.float64)
    sd = _side(K, c, stats_c=sc)
    sd.stats = rows[c:]                      # the second edge of the run: its first channel inside the rows
    K._compact_stats(sd)
    assert sd.stats_c == 0 and sd.stats.numel() == K.R * 2 * c
    got = sd.stats.view(K.R, 2, c)
    want = rows.view(K.R, 2, sc)[:, :, c:2 * c]
    assert torch.equal(got, want) and got.data_ptr() != rows.data_ptr()


@pytest.mark.parametrize("tiles,nk,want", [(144, 36, 3), (72, 18, 3), (72, 16, 2), (144, 16, 1), (288, 18, 1), (36, 36, 4)])
def test_split_k_share_count_follows_the_rounds_model(tiles, nk, want):
    """conv_g4_launch's choice restated: S in 1..4 (S * 6 <= K-tiles) minimising rounds-of-256(S * tiles) / S + 0.05 S, only for grids
    of fewer tiles than CUs with >= 12 K-tiles (csrc/conv_g4.hip; the GPU test checks the launches' own report against the same numbers)."""
    best, cost = 1, (tiles + 255) // 256 + 0.05
    if tiles < 256 and nk >= 12:
        for s in (2, 3, 4):
            if s * 6 <= nk:
                c = ((tiles * s + 255) // 256) / s + 0.05 * s
                if c < cost - 1e-9:
                    best, cost = s, c
    assert best == want
