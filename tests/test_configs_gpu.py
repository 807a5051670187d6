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


@pytest.mark.parametrize("batch,size", [(16, 384), (32, 384), (8, 512)])   # (16, 384) is the bench line's own workload
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


def test_supernet_at_its_benchmark_size_trains_under_the_graph():
    """BASELINE config 5 at its real per-GPU size: the MixedOp supernet model_search_interact.Network, C = 32,
    L = 16, 384 x 384, batch 8, bf16 -- both passes of train_with_alpha (core/function.py:485-621) through SearchStep, each captured
    as its own hipGraph; the parity of the supernet is pinned at the golden size (tests/test_ops_gpu.py), here the kernels, tile
    choices and memory plan of the benchmark shape run, replay and train."""
    from types import SimpleNamespace as NS
    from npp_amd.criterion import Criterion_par, Criterion_pose
    from npp_amd.model_augment import set_compute_dtype
    from npp_amd.model_search_interact import Network as SearchNetwork
    from npp_amd.optim import FusedAdam
    from npp_amd.synth import synth_batch
    from npp_amd.train_step import SearchStep
    dev = torch.device("cuda:0")
    set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(0)
        net = SearchNetwork(NS(DATASET=NS(NUM_CLASSES=20, NUM_JOINTS=16), SEARCH=NS(LAYERS=16, INIT_CHANNELS=32),
                               MODEL=NS(DECONV_WITH_BIAS=False, HEAD='PSP', REFINE_LAYERS=1))).to(dev).train()
        cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
        arch = list(net.arch_parameters())
        arch_ids = {id(a) for a in arch}
        weights = [p for p in list(net.parameters()) + list(cp.parameters()) + list(cq.parameters()) if id(p) not in arch_ids]
        before = [a.detach().clone() for a in arch]
        step = SearchStep(net, cp, cq, FusedAdam(weights, lr=1e-4), FusedAdam(arch, lr=3e-3, betas=(0.5, 0.999), weight_decay=0.001),
                          graph=True, warmup=1)

        def batch(seed):
            images, lpar, lpose, _ = synth_batch(8, 384, seed=seed)
            return (torch.from_numpy(images).to(dev), [torch.from_numpy(a).to(dev) for a in lpar],
                    [torch.from_numpy(a[:, :-1].copy()).to(dev) for a in lpose])
        b1, b2 = batch(0), batch(1)
        losses = [tuple(float(x) for x in step(b1, b2)) for _ in range(4)]
        torch.cuda.synchronize()
        assert step.weights_pass.graphed and step.alpha_pass(False).graphed, "a pass was not captured"
        assert all(l == l and abs(l) < 1e4 for pair in losses for l in pair), losses
        assert losses[-1][0] < losses[0][0], losses                 # the same batches four times: the weights pass makes progress
        assert all(torch.isfinite(p).all() for p in net.parameters())
        assert all(float((a - b).abs().max()) > 0 for a, b in zip(arch, before))      # the architecture pass moved alpha / beta
        assert torch.cuda.max_memory_allocated() < 120e9
    finally:
        set_compute_dtype(torch.float32)
