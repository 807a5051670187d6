"""world_size-2 gloo test of the bucketed gradient reducer: N ranks x batch B must produce the gradient of
1 rank x batch N*B (mean loss), with statically skipped never-used parameters and several buckets."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(16, 64)
        self.b = torch.nn.Linear(64, 64)
        self.unused = torch.nn.Linear(4, 4)     # never touched by forward (like SE_Block.bn at stride 1)
        self.c = torch.nn.Linear(64, 3)

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _worker(rank, world, port, out, overlap=True):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from npp_amd.ddp import GradReducer
    torch.manual_seed(123 + rank)          # different init per rank: the reducer must broadcast rank 0's
    m = _Toy()
    red = GradReducer(m, bucket_mb=0.01, skip={"unused.weight", "unused.bias"}, overlap=overlap)
    assert len(red.buckets) >= 3
    torch.manual_seed(7)
    xs = torch.randn(world * 5, 16)
    ys = torch.randn(world * 5, 3)
    for step in range(2):
        m.zero_grad()
        loss = ((m(xs[rank * 5:(rank + 1) * 5]) - ys[rank * 5:(rank + 1) * 5]) ** 2).mean()
        loss.backward()
        red.finish()
    torch.save({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, f"{out}/g{rank}.pt")
    if rank == 0:
        torch.save(m.state_dict(), f"{out}/sd.pt")
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("overlap", [True, False])
def test_grad_reducer_matches_single_process(tmp_path, overlap):
    world, port = 2, 29500 + os.getpid() % 1000 + (0 if overlap else 1000)
    mp.spawn(_worker, args=(world, port, str(tmp_path), overlap), nprocs=world, join=True)
    sd = torch.load(f"{tmp_path}/sd.pt")
    m = _Toy()
    m.load_state_dict(sd)
    torch.manual_seed(7)
    xs = torch.randn(world * 5, 16)
    ys = torch.randn(world * 5, 3)
    ((m(xs) - ys) ** 2).mean().backward()
    for r in range(world):
        g = torch.load(f"{tmp_path}/g{r}.pt")
        assert "unused.weight" not in g
        for k, p in m.named_parameters():
            if p.grad is not None:
                assert torch.allclose(g[k], p.grad, atol=1e-6), k


class _ToyConv(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.k3 = torch.nn.Conv2d(2, 6, 3, padding=1)
        self.dw = torch.nn.Conv2d(6, 6, 3, padding=1, groups=6)      # depthwise: [6, 1, 3, 3] -- not a dense KxK weight
        self.p1 = torch.nn.Conv2d(6, 3, 1)
        self.k5 = torch.nn.Conv2d(3, 2, 5, padding=2)

    def forward(self, x):
        return self.k5(torch.relu(self.p1(self.dw(torch.relu(self.k3(x))))))


def _tail_worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from npp_amd.ddp import GradReducer
    torch.manual_seed(5 + rank)
    m = _ToyConv()
    red = GradReducer(m, bucket_mb=0.0001, overlap="tail")
    kinds = [b.kind for b in red.buckets]
    assert kinds == sorted(kinds) and kinds.count("K") >= 2 and kinds.count("O") >= 2, kinds      # per-kind runs of buckets, K first
    k_params = {id(p) for b in red.buckets if b.kind == "K" for p in b.params}
    assert k_params == {id(m.k3.weight), id(m.k5.weight)}
    torch.manual_seed(9)
    xs, ys = torch.randn(world * 3, 2, 6, 6), torch.randn(world * 3, 2, 6, 6)
    for step in range(2):
        m.zero_grad()
        ((m(xs[rank * 3:(rank + 1) * 3]) - ys[rank * 3:(rank + 1) * 3]) ** 2).mean().backward()
        red.launch_kind("K")          # (where TrainStep has just finished the KxK group of its weight-gradient tail)
        assert all(b.launched == (b.kind == "K") for b in red.buckets)
        red.finish()
    torch.save({k: p.grad.clone() for k, p in m.named_parameters()}, f"{out}/g{rank}.pt")
    if rank == 0:
        torch.save(m.state_dict(), f"{out}/sd.pt")
    dist.destroy_process_group()


def test_tail_overlap_reducer_buckets_per_kind_and_staged_launch(tmp_path):
    """GradReducer(overlap="tail"): dense KxK conv weights get buckets of their own, launch_kind("K") reduces exactly those,
    finish() the rest; 2 ranks x 3 samples == 1 rank x 6 samples."""
    world, port = 2, 31500 + os.getpid() % 1000
    mp.spawn(_tail_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    m = _ToyConv()
    m.load_state_dict(torch.load(f"{tmp_path}/sd.pt"))
    torch.manual_seed(9)
    xs, ys = torch.randn(world * 3, 2, 6, 6), torch.randn(world * 3, 2, 6, 6)
    ((m(xs) - ys) ** 2).mean().backward()
    for r in range(world):
        g = torch.load(f"{tmp_path}/g{r}.pt")
        for k, p in m.named_parameters():
            assert torch.allclose(g[k], p.grad, atol=1e-6), k


def test_unused_parameter_names_cover_se_block_bn():
    sys.path.insert(0, REPO)
    from types import SimpleNamespace as NS
    from npp_amd.model_augment import Network
    from npp_amd.ddp import unused_parameter_names
    cfg = NS(DATASET=NS(NUM_CLASSES=20, NUM_JOINTS=16), TRAIN=NS(LAYERS=16, INIT_CHANNELS=16),
             MODEL=NS(DECONV_WITH_BIAS=False, HEAD='PSP', REFINE_LAYERS=1))
    net = Network(cfg)
    names = unused_parameter_names(net)
    from helpers import load_golden
    g = load_golden("tiny_net.npz")
    assert names == set(str(k) for k in g["train/no_grad_keys"])   # the reference's never-produced gradients
    assert len(names) == 116


def _agree_worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from npp_amd.train_step import TrainStep
    ts = TrainStep.__new__(TrainStep)           # only the agreement helper is exercised here (no GPU in this test)
    got = [ts._all_ranks_ok(True), ts._all_ranks_ok(rank != 1), ts._all_ranks_ok(False)]
    torch.save(got, f"{out}/a{rank}.pt")
    dist.destroy_process_group()


def test_capture_agreement_is_unanimous(tmp_path):
    """TrainStep's "captured OK" vote (train_step.py:_all_ranks_ok): one failing rank sends EVERY rank to the eager path --
    a rank replaying the lockstep collective order next to a rank issuing the single-stream order would hang."""
    world, port = 2, 31500 + os.getpid() % 1000
    mp.spawn(_agree_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert torch.load(f"{tmp_path}/a{r}.pt") == [True, False, False]
