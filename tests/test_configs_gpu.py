"""BASELINE.json configurations at their full per-GPU shapes on one MI355X: config 3's batch 32 per GPU at
384 x 384 and config 4's batch 8 at 512 x 512, C = 64, bf16, through the captured TrainStep -- the kernels, tile choices and
memory plan those shapes select (h3 / g8 / wgrad splits change with the pixel count) run, are replayed, and train.  Numerical
parity of the same shapes is pinned elsewhere (tests/test_ops_gpu.py: full 384 / 512 networks against the reference)."""
import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("batch,size", [(32, 384), (8, 512)])
def test_full_size_configuration_trains_under_the_graph(batch, size):
    import bench
    from npp_amd.criterion import Criterion_par, Criterion_pose
    from npp_amd.model_augment import Network, set_compute_dtype
    from npp_amd.optim import FusedAdam
    from npp_amd.synth import synth_batch
    from npp_amd.train_step import TrainStep
    dev = torch.device("cuda:0")
    set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(0)
        net = Network(bench.cfg_ns()).to(dev).train()
        cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
        opt = FusedAdam(list(net.parameters()) + list(cp.parameters()) + list(cq.parameters()), lr=1e-4)
        step = TrainStep(net, cp, cq, opt, graph=True, warmup=1)
        images, lpar, lpose, _ = synth_batch(batch, size, seed=0)
        images = torch.from_numpy(images).to(dev)
        lpar = [torch.from_numpy(a).to(dev) for a in lpar]
        lpose = [torch.from_numpy(a[:, :-1].copy()).to(dev) for a in lpose]
        losses = [float(step(images, lpar, lpose).detach()) for _ in range(5)]
        torch.cuda.synchronize()
        assert step.graphed, "the step was not captured"
        assert all(l == l and abs(l) < 1e4 for l in losses), losses
        assert losses[-1] < losses[0], losses                     # the same batch five times: Adam must make progress
        assert all(torch.isfinite(p).all() for p in net.parameters())
        assert torch.cuda.max_memory_allocated() < 120e9          # the memory plan of DESIGN.md section 3 (288 GB per GPU)
    finally:
        set_compute_dtype(torch.float32)
