"""npp_amd.auto_graph: hipGraph replay of Network.forward + backward behind an UNCHANGED launcher loop (model(images), criteria,
zero_grad, backward, torch.optim.Adam.step -- core/function.py:87-107).  The replayed forward / backward must give what the
eager ones give from identical state; eval, other batch shapes and gradient accumulation must keep working."""
import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

pytestmark = pytest.mark.gpu


@pytest.fixture()
def auto_on():
    from npp_amd import _ops as K, auto_graph
    old = auto_graph.ENABLED
    auto_graph.ENABLED = True
    yield auto_graph
    auto_graph.ENABLED = old
    K.GRAPH_TOPOLOGY = False


def _setup(dev, seed=0):
    import test_train_step_gpu as T
    from npp_amd.criterion import Criterion_par, Criterion_pose
    from npp_amd.model_augment import Network, set_compute_dtype
    set_compute_dtype(torch.float32)
    torch.manual_seed(seed)
    net = Network(T._cfg(int(os.environ.get("AUTO_TEST_C", "8")))).to(dev).train()
    cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
    return net, cp, cq


def _loss(net, cp, cq, batch):
    im, lpar, lpose, _w = batch
    output_pose, output_par = net(im)
    return (cq(output_par, lpar).unsqueeze(0) + cp(output_pose, lpose).unsqueeze(0)).mean()


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _copy_state(src, dst):
    with torch.no_grad():
        for a, b in zip(list(src.parameters()) + list(src.buffers()), list(dst.parameters()) + list(dst.buffers())):
            b.copy_(a)


def test_replayed_forward_backward_equal_the_eager_ones(auto_on):
    import test_train_step_gpu as T
    dev = torch.device("cuda:0")
    net_a, cp_a, cq_a = _setup(dev)          # replayed (after two eager calls)
    net_b, cp_b, cq_b = _setup(dev)          # eager twin
    net_b._auto_graph_off = True
    batch = T._batch(2, 64, 3, dev)
    opt = torch.optim.Adam(list(net_a.parameters()) + list(cp_a.parameters()) + list(cq_a.parameters()), lr=1e-3)
    for it in range(5):
        _copy_state(net_a, net_b)
        with torch.no_grad():
            for a, b in zip(list(cp_a.parameters()) + list(cq_a.parameters()), list(cp_b.parameters()) + list(cq_b.parameters())):
                b.copy_(a)
        opt.zero_grad()
        la = _loss(net_a, cp_a, cq_a, batch)
        la.backward()
        for p in list(net_b.parameters()) + list(cp_b.parameters()) + list(cq_b.parameters()):
            p.grad = None
        lb = _loss(net_b, cp_b, cq_b, batch)
        lb.backward()
        torch.cuda.synchronize()
        assert abs(float(la.detach()) - float(lb.detach())) < 2e-4 * abs(float(lb.detach())), (it, float(la.detach()), float(lb.detach()))
        pa, pb = dict(net_a.named_parameters()), dict(net_b.named_parameters())
        worst = 0.0
        for k in pa:
            assert (pa[k].grad is None) == (pb[k].grad is None), k
            if pa[k].grad is not None and float(pb[k].grad.norm()) > 1e-6:
                worst = max(worst, rel_l2(pa[k].grad, pb[k].grad))
        assert worst < 5e-2, (it, worst)        # (OHEM / arg-max flips from the float atomics' order: tests/test_ops_gpu.py)
        # running statistics moved identically
        for (ka, ba), (kb, bb) in zip(net_a.named_buffers(), net_b.named_buffers()):
            if ba.is_floating_point():
                assert rel_l2(ba, bb) < 1e-4, ka
        opt.step()
    assert net_a._auto is not None and net_a._auto.graph is not None, "the forward was never captured"
    assert net_b._auto is None or net_b._auto.graph is None


def test_eval_other_shapes_and_accumulation(auto_on):
    import test_train_step_gpu as T
    from npp_amd.model_augment import Network
    dev = torch.device("cuda:0")
    net, cp, cq = _setup(dev, seed=1)
    batch = T._batch(2, 64, 3, dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = _loss(net, cp, cq, batch)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert net._auto.graph is not None and losses[-1] < losses[0] and all(l == l for l in losses)
    # a batch of another shape runs eagerly, the graph stays
    other = T._batch(1, 96, 4, dev)
    opt.zero_grad()
    _loss(net, cp, cq, other).backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    opt.zero_grad()
    # gradient accumulation: a second backward before zero_grad must ADD -- two DIFFERENT batches of the captured shape, against
    # the sum of their separately computed gradients (the parameters do not move in between; ADVICE r2: the replay used to
    # overwrite the first gradient, giving 2*g2)
    batch2 = T._batch(2, 64, 7, dev)
    grads = []
    for bt in (batch, batch2):
        opt.zero_grad()
        _loss(net, cp, cq, bt).backward()
        grads.append({k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    opt.zero_grad()
    _loss(net, cp, cq, batch).backward()
    _loss(net, cp, cq, batch2).backward()
    torch.cuda.synchronize()
    params = dict(net.named_parameters())
    for k in ("stem0.0.weight", "cells1.3.preprocess1.net.1.weight", "pose_head.1.4.weight", "par_layer.1.weight"):
        want = grads[0][k] + grads[1][k]
        assert rel_l2(params[k].grad, want) < 5e-2, (k, rel_l2(params[k].grad, want))
        assert rel_l2(grads[0][k], grads[1][k]) > 0.2, k          # (the two batches do give different gradients)
    # eval after replays: equals a fresh network loaded from the state dict
    net.eval()
    with torch.no_grad():
        pose, par = net(batch[0])
    fresh = Network(T._cfg(8)).to(dev)
    fresh.load_state_dict(net.state_dict())
    fresh.eval()
    with torch.no_grad():
        pose2, par2 = fresh(batch[0])
    assert rel_l2(par[-1][0], par2[-1][0]) < 1e-5 and rel_l2(pose[-1][0], pose2[-1][0]) < 1e-5


def test_search_supernet_is_replayed_too(auto_on):
    """model_search_interact.Network behind the same switch: the loop body of train_with_alpha calls model(input) twice per
    iteration (train batch, mini-loader batch) with every parameter requiring grad -- one graph serves both calls."""
    import test_train_step_gpu as T
    dev = torch.device("cuda:0")
    net, cp, cq, weights = T._search_setup(dev)
    arch = list(net.arch_parameters())
    opt = torch.optim.Adam(weights, lr=1e-3)
    a_opt = torch.optim.Adam(arch, lr=3e-3, betas=(0.5, 0.999), weight_decay=1e-3)
    b1, b2 = T._batch(2, 64, 3, dev), T._batch(2, 64, 4, dev)
    seen = []
    for it in range(4):
        opt.zero_grad()
        loss = _loss(net, cp, cq, b1)
        loss.backward()
        opt.step()
        a_opt.zero_grad()
        loss2 = _loss(net, cp, cq, b2) + 2 * net.loss_entropy()
        loss2.backward()
        a_opt.step()
        seen.append((float(loss.detach()), float(loss2.detach())))
    assert net._auto is not None and net._auto.graph is not None
    assert all(a == a and b == b for a, b in seen) and seen[-1][0] < seen[0][0], seen
    assert all(a.grad is not None and torch.isfinite(a.grad).all() and float(a.grad.abs().sum()) > 0 for a in arch)
