"""The one-shot peer-to-peer SyncBatchNorm exchange (csrc/p2p.hip, npp_amd/comm.py:enable_p2p) with two real processes.

Both ranks share the one GPU of the test box (hipIpc maps a buffer of the other PROCESS just the same; the peer stores then go
through one L2 instead of xGMI): mailbox set-up over a gloo side channel, exchanges of every size, hipGraph replays (the sequence
counter lives on the device), two channels on two streams, a peer that never shows up (bounded poll, error word, NaN sums, the
collective health check), and the
SyncBatchNorm network of tests/test_syncbn_gpu.py with this transport against the same goldens."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

pytestmark = pytest.mark.gpu


def _vec(rank, it, n):
    g = np.random.default_rng(1000 * it + rank)
    return g.integers(-1000, 1000, n).astype(np.float64) / 8.0      # sums are exact in f64 whatever the order


def _worker(rank, world, port, out, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["NPP_P2P_TIMEOUT_MS"] = "4000"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from npp_amd import comm
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    res = {"enabled": bool(comm.enable_p2p(None, channels=2))}  # (two channels: the default stream and the graph's side stream)
    if not res["enabled"]:
        torch.save(res, f"{out}.{rank}")
        dist.destroy_process_group()
        return
    cap = comm._p2p["cap"]
    bad = []
    if mode == "basic":
        sizes = [1, 2, 7, 64, 1000, 4097, cap, 3, cap - 1, 512] * 3
        for it, n in enumerate(sizes):
            v = torch.from_numpy(_vec(rank, it, n)).to(dev)
            assert comm.p2p_exchange(v, None)
            want = sum(_vec(r, it, n) for r in range(world))
            if not np.array_equal(v.cpu().numpy(), want):
                bad.append(("eager", it, n))
        big = torch.from_numpy(_vec(rank, 999, 2 * cap + 5)).to(dev)      # longer than a mailbox slot: three pieces
        assert comm.p2p_exchange(big, None)
        res["long_vector_ok"] = bool(np.array_equal(big.cpu().numpy(), sum(_vec(r, 999, 2 * cap + 5) for r in range(world))))
        # hipGraph: three exchanges per replay on a side stream (a channel of its own), static input refreshed between replays
        st = torch.cuda.Stream()
        n = 777
        x = torch.zeros(3, n, dtype=torch.float64, device=dev)
        y = torch.zeros_like(x)
        with torch.cuda.stream(st):
            y.copy_(x)
            for k in range(3):
                assert comm.p2p_exchange(y[k], None)       # (warm-up: also assigns the stream its channel outside the capture)
            st.synchronize()
            dist.barrier()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                y.copy_(x)
                for k in range(3):
                    assert comm.p2p_exchange(y[k], None)
        for rep in range(6):
            x.copy_(torch.from_numpy(np.stack([_vec(rank, 100 + 3 * rep + k, n) for k in range(3)])).to(dev))
            torch.cuda.synchronize()
            gr.replay()
            torch.cuda.synchronize()
            want = np.stack([sum(_vec(r, 100 + 3 * rep + k, n) for r in range(world)) for k in range(3)])
            if not np.array_equal(y.cpu().numpy(), want):
                bad.append(("graph", rep))
        # slab form: two segments of replica slabs (16 x 2c and 16 x 3c), local sums out as floats, the world's sum in replica 0,
        # the other replicas of the second segment zeroed
        c1, c2, R = 40, 24, 16
        g = np.random.default_rng(7 + rank)
        s1 = g.integers(-50, 50, (R, 2 * c1)).astype(np.float64)
        s2 = g.integers(-50, 50, (R, 3 * c2)).astype(np.float64)
        t1, t2 = torch.from_numpy(s1.copy()).to(dev).reshape(-1), torch.from_numpy(s2.copy()).to(dev).reshape(-1)
        o = [torch.full((n_,), -1.0, device=dev) for n_ in (c1, c1, c2, c2, c2, c2)]
        ok_ = comm.p2p_exchange_slabs([(t1, 2 * c1, R, c1, (o[0], None, o[1], None), False),
                                       (t2, 3 * c2, R, c2, (o[2], o[3], o[4], o[5]), True)], None)
        torch.cuda.synchronize()
        loc1, loc2 = s1.sum(0), s2.sum(0)
        all1 = sum(np.random.default_rng(7 + r).integers(-50, 50, (R, 2 * c1)).astype(np.float64).sum(0) for r in range(world))
        gens = [np.random.default_rng(7 + r) for r in range(world)]
        all2 = 0
        for gg in gens:
            gg.integers(-50, 50, (R, 2 * c1))
            all2 = all2 + gg.integers(-50, 50, (R, 3 * c2)).astype(np.float64).sum(0)
        r1, r2 = t1.cpu().numpy().reshape(R, -1), t2.cpu().numpy().reshape(R, -1)
        res["slabs_ok"] = bool(ok_ and np.array_equal(r1[0], all1) and np.array_equal(r1[1:], s1[1:])      # (not zeroed: untouched)
                               and np.array_equal(r2[0], all2) and not r2[1:].any()
                               and np.array_equal(o[0].cpu().numpy(), loc1[:c1].astype(np.float32))
                               and np.array_equal(o[1].cpu().numpy(), loc1[c1:].astype(np.float32))
                               and np.array_equal(o[2].cpu().numpy(), loc2[:c2].astype(np.float32))
                               and np.array_equal(o[3].cpu().numpy(), loc2[:c2].astype(np.float32))
                               and np.array_equal(o[4].cpu().numpy(), loc2[c2:2 * c2].astype(np.float32))
                               and np.array_equal(o[5].cpu().numpy(), loc2[2 * c2:].astype(np.float32)))
        res["exchanges"] = comm._p2p["count"]
    elif mode == "timeout":
        # rank 1 never joins the second exchange: rank 0's poll gives up after NPP_P2P_TIMEOUT_MS (4 s here) and reports it
        v = torch.ones(8, dtype=torch.float64, device=dev)
        assert comm.p2p_exchange(v, None)
        torch.cuda.synchronize()
        res["first_ok"] = bool(comm.p2p_ok()) and float(v[0]) == world
        if rank == 0:
            w = torch.ones(8, dtype=torch.float64, device=dev)
            assert comm.p2p_exchange(w, None)
            torch.cuda.synchronize()
            res["second_reported"] = not comm.p2p_ok()
            res["second_is_nan"] = bool(torch.isnan(w).all())      # a failed exchange never hands back local sums
            import time
            t0 = time.time()
            for _ in range(50):                         # a dead channel does not wait again: 50 more exchanges take no time
                assert comm.p2p_exchange(w, None)
            torch.cuda.synchronize()
            res["dead_channel_is_fast"] = time.time() - t0 < 2.0
            res["dead_channel_is_nan"] = bool(torch.isnan(w).all())
        # the collective health check TrainStep runs every few steps: BOTH ranks raise, although only rank 0 saw the time-out
        try:
            comm.p2p_check(what="the test")
            res["check_raised"] = False
        except RuntimeError:
            res["check_raised"] = True
    res["bad"] = bad
    res["mode"] = comm._p2p.get("mode")
    res["ok"] = bool(comm.p2p_ok()) if mode != "timeout" else True
    torch.save(res, f"{out}.{rank}")
    dist.barrier()
    comm.disable_p2p()
    dist.destroy_process_group()


def _run(mode, tmp_path, port):
    out = str(tmp_path / "p2p")
    mp.spawn(_worker, args=(2, port, out, mode), nprocs=2, join=True)
    return [torch.load(f"{out}.{r}") for r in range(2)]


def test_two_processes_exchange_through_ipc_mailboxes(tmp_path):
    res = _run("basic", tmp_path, 29671)
    if not all(r["enabled"] for r in res):
        assert not any(r["enabled"] for r in res), "the ranks must agree on the transport"
        pytest.skip("this runtime refuses hipIpc between two processes of one device")
    for r in res:
        assert r["bad"] == [] and r["ok"] and r["long_vector_ok"] and r["slabs_ok"]
        assert r["mode"] in ("relaxed", "fenced")      # the 2000-exchange acceptance test of enable_p2p passed in this mode
        assert r["exchanges"] == 30 + 3 + 3 + 3 + 1      # (replays do not pass through the host counter)


def test_a_missing_peer_is_reported_not_waited_for_forever(tmp_path):
    res = _run("timeout", tmp_path, 29673)
    if not all(r["enabled"] for r in res):
        pytest.skip("this runtime refuses hipIpc between two processes of one device")
    assert all(r["first_ok"] for r in res)
    assert res[0]["second_reported"] and res[0]["dead_channel_is_fast"]
    assert res[0]["second_is_nan"] and res[0]["dead_channel_is_nan"]
    assert all(r["check_raised"] for r in res), "the periodic health check must stop EVERY rank"
