"""npp_amd.train_step.TrainStep (the loop body of core/function.py:72-107): the hipGraph-replayed step must follow the same
trajectory as the eager step -- same losses, same parameters after several optimiser steps -- including a learning-rate
change after capture (augment_lip_sync.py:213,249) and a batch of another shape (runs eagerly)."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

pytestmark = pytest.mark.gpu


def _cfg(C):
    from types import SimpleNamespace as NS
    return NS(DATASET=NS(NUM_CLASSES=20, NUM_JOINTS=16), TRAIN=NS(LAYERS=16, INIT_CHANNELS=C),
              MODEL=NS(DECONV_WITH_BIAS=False, HEAD='PSP', REFINE_LAYERS=1))


def _batch(n, size, seed, dev):
    from npp_amd.synth import synth_batch
    images, lpar, lpose, meta = synth_batch(n, size, seed=seed)
    return (torch.from_numpy(images).to(dev), [torch.from_numpy(a).to(dev) for a in lpar],
            [torch.from_numpy(a[:, :-1].copy()).to(dev) for a in lpose], torch.from_numpy(meta["pose_weight"]).to(dev))


def _make(dev, graph, state=None):
    from npp_amd.criterion import Criterion_par, Criterion_pose
    from npp_amd.model_augment import Network, set_compute_dtype
    from npp_amd.optim import FusedAdam
    from npp_amd.train_step import TrainStep
    set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    net = Network(_cfg(8)).to(dev).train()
    if state is not None:
        net.load_state_dict(state)
    cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
    opt = FusedAdam(list(net.parameters()) + list(cp.parameters()) + list(cq.parameters()), lr=1e-3)
    return net, opt, TrainStep(net, cp, cq, opt, graph=graph, warmup=1)


def _sync_state(src, dst):
    """Parameters, buffers and Adam state of (net, criteria, optimizer) `src` -> `dst`, in place."""
    (net_s, cp_s, cq_s, opt_s), (net_d, cp_d, cq_d, opt_d) = src, dst
    with torch.no_grad():
        for ms, md in ((net_s, net_d), (cp_s, cp_d), (cq_s, cq_d)):
            for a, b in zip(list(ms.parameters()) + list(ms.buffers()), list(md.parameters()) + list(md.buffers())):
                b.copy_(a)
        for gs, gd in zip(opt_s.param_groups, opt_d.param_groups):
            for a, b in zip(gs["params"], gd["params"]):
                if "exp_avg" in opt_s.state.get(a, {}):
                    opt_d._state_for(b)
                    opt_d.state[b]["exp_avg"].copy_(opt_s.state[a]["exp_avg"])
                    opt_d.state[b]["exp_avg_sq"].copy_(opt_s.state[a]["exp_avg_sq"])
        if opt_s._step is not None and opt_d._step is not None:
            opt_d._step.copy_(opt_s._step)


def test_graphed_step_equals_the_eager_step():
    """Every step starts from identical state (copied graphed -> eager), so the two runs can only differ by float-atomic
    summation order: equal losses, and equal Adam updates except where a near-zero gradient's sign is noise."""
    dev = torch.device("cuda:0")
    net_e, opt_e, step_e = _make(dev, graph=False)
    net_g, opt_g, step_g = _make(dev, graph=True)
    eager = (net_e, step_e.criterion_pose, step_e.criterion_par, opt_e)
    graphed = (net_g, step_g.criterion_pose, step_g.criterion_par, opt_g)
    batches = [_batch(2, 128, s, dev) for s in range(6)]
    short = _batch(2, 96, 99, dev)
    pe, pg = dict(net_e.named_parameters()), dict(net_g.named_parameters())
    lr = 1e-3

    def both(batch, i):
        if i > 0:
            _sync_state(graphed, eager)
        before = {k: v.detach().clone() for k, v in pg.items()}
        im, lpar, lpose, _ = batch
        le = float(step_e(im, list(lpar), list(lpose)).detach())
        lg = float(step_g(im, list(lpar), list(lpose)).detach())
        assert abs(le - lg) <= 2e-5 * abs(le), (i, le, lg)
        off = tot = 0
        biggest = 0.0
        for k in pg:
            de, dg = pe[k].detach() - before[k], pg[k].detach() - before[k]
            off += int(((de - dg).abs() > 0.02 * lr).sum())
            tot += de.numel()
            biggest = max(biggest, float(dg.abs().max()))
        assert off <= 2e-3 * tot, (i, off, tot)
        return biggest

    for i, batch in enumerate(batches):
        if i == 4:                       # MultiStepLR milestone after capture (augment_lip_sync.py:213,249)
            lr = 1e-4
            for opt in (opt_e, opt_g):
                for grp in opt.param_groups:
                    grp["lr"] = lr
        biggest = both(batch, i)
        assert 0.2 * lr < biggest < 3.5 * lr, (i, biggest, lr)      # the replayed graph steps with the CURRENT lr
        if i == 2:                       # a batch of another shape in between: runs eagerly, the graph survives it
            both(short, 100)
    assert step_g.graphed and not step_e.graphed
    assert opt_g.device_step_count() == opt_e.device_step_count() == 7


def test_pose_weight_argument_and_static_inputs():
    dev = torch.device("cuda:0")
    from npp_amd.criterion import Criterion_par, Criterion_pose
    from npp_amd.model_augment import Network, set_compute_dtype
    from npp_amd.optim import FusedAdam
    from npp_amd.train_step import TrainStep
    set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    net = Network(_cfg(8)).to(dev).train()
    cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
    opt = FusedAdam(list(net.parameters()) + list(cp.parameters()) + list(cq.parameters()), lr=1e-3)
    step = TrainStep(net, cp, cq, opt, warmup=1)
    im, lpar, lpose, w = _batch(2, 64, 3, dev)
    l0 = float(step(im, lpar, lpose, w).detach())
    l1 = float(step(im, lpar, lpose, w).detach())
    l2 = float(step(im, lpar, lpose, w).detach())
    assert step.graphed and len(step.static_inputs) == 6
    assert np.isfinite([l0, l1, l2]).all() and l2 < l0


def test_eval_after_graph_replays_uses_the_updated_weights():
    """The replayed graph updates the parameters without any host-side version bump (FusedAdam's increment_version ran once,
    at capture): an eval forward after N replays must still run on operand images of the CURRENT weights, i.e. equal the eval
    output of a freshly built model loaded from state_dict() -- and stay right for a second train / eval round."""
    from npp_amd.model_augment import Network
    dev = torch.device("cuda:0")
    net, opt, step = _make(dev, graph=True)
    for grp in opt.param_groups:
        grp["lr"] = 2e-2                   # large steps: stale images would be far off
    im, lpar, lpose, _ = _batch(2, 64, 5, dev)

    def eval_out(model):
        model.eval()
        with torch.no_grad():
            pose, par = model(im)
        model.train()
        return pose[1][0].float().clone(), par[1][0].float().clone()

    for rnd in range(2):
        for _ in range(4):
            step(im, list(lpar), list(lpose))
        assert step.graphed
        got = eval_out(net)
        fresh = Network(_cfg(8)).to(dev)
        fresh.load_state_dict(net.state_dict())
        want = eval_out(fresh)
        for a, b in zip(got, want):
            assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max())), rnd


def test_capture_failure_falls_back_to_eager():
    """A hipGraph capture that is invalidated half way leaves the capture stream current and the side streams stuck in
    capture mode (tools/capture_recover.py); TrainStep must restore the stream, take fresh side streams and keep stepping
    eagerly.  Run in a subprocess: torch's own capture bookkeeping stays odd after such a failure."""
    import subprocess
    env = dict(os.environ, NPP_TEST_FAIL_CAPTURE="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "capture_fail_worker.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "FALLBACK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert "capture failed" in r.stderr


def _search_setup(dev, seed=0):
    from types import SimpleNamespace as NS
    from npp_amd.criterion import Criterion_par, Criterion_pose
    from npp_amd.model_augment import set_compute_dtype
    from npp_amd.model_search_interact import Network
    set_compute_dtype(torch.float32)
    torch.manual_seed(seed)
    cfg = NS(DATASET=NS(NUM_CLASSES=20, NUM_JOINTS=16), SEARCH=NS(LAYERS=16, INIT_CHANNELS=8),
             MODEL=NS(DECONV_WITH_BIAS=False, HEAD='PSP', REFINE_LAYERS=1))
    net = Network(cfg).to(dev).train()
    with torch.no_grad():
        for i, a in enumerate(net.arch_parameters()):      # non-uniform architecture weights
            a.copy_(torch.linspace(-1, 1, a.numel(), device=dev).reshape(a.shape) * (0.5 + 0.1 * i))
    cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
    arch_ids = {id(a) for a in net.arch_parameters()}
    weights = [p for p in net.parameters() if id(p) not in arch_ids] + list(cp.parameters()) + list(cq.parameters())
    return net, cp, cq, weights


def test_search_step_equals_the_reference_loop_body():
    """SearchStep (train_with_alpha, core/function.py:485-621) against the loop body written out as the reference has it --
    plain backward of EVERY parameter in both passes -- on a twin network brought to identical state before each pass: the
    weights pass must produce the twin's weight gradients and loss, the alpha pass the twin's architecture gradients and loss
    (second iteration with the entropy term, epoch > 70); freezing the other pass's parameters may change nothing but the work
    done.  (Whole trajectories are not compared: from any 1e-6 difference this train-mode network diverges, see
    tests/test_ops_gpu.py on bf16.)"""
    from npp_amd.optim import FusedAdam
    from npp_amd.train_step import SearchStep
    from helpers import rel_l2
    dev = torch.device("cuda:0")
    net_a, cp_a, cq_a, w_a = _search_setup(dev)
    net_b, cp_b, cq_b, w_b = _search_setup(dev)
    opt_a = FusedAdam(w_a, lr=1e-3)
    aopt_a = FusedAdam(net_a.arch_parameters(), lr=3e-3, betas=(0.5, 0.999), weight_decay=0.001)
    step = SearchStep(net_a, cp_a, cq_a, opt_a, aopt_a, graph=False)
    arch_a, arch_b = list(net_a.arch_parameters()), list(net_b.arch_parameters())

    def sync():
        with torch.no_grad():
            for ma, mb in ((net_a, net_b), (cp_a, cp_b), (cq_a, cq_b)):
                for x, y in zip(list(ma.parameters()) + list(ma.buffers()), list(mb.parameters()) + list(mb.buffers())):
                    y.copy_(x)
                for y in mb.parameters():
                    y.grad = None

    def close(ga, gb, what):
        assert ga is not None and gb is not None, what
        e = rel_l2(ga.detach().float().cpu().numpy(), gb.detach().float().cpu().numpy())
        assert e < 2e-3 or float(gb.abs().max()) < 1e-7, (what, e)

    for it in range(2):
        b1, b2 = _batch(2, 64, 10 + it, dev), _batch(2, 64, 20 + it, dev)
        entropy = it == 1
        # ---- weights pass (core/function.py:499-531) -------------------------------------------------------------------
        sync()
        im, lpar, lpose, pw = b1
        po, pa = net_b(im)
        loss1 = (cq_b(pa, lpar).unsqueeze(0) + cp_b(po, lpose, target_weight=pw).unsqueeze(0)).mean()
        loss1.backward()
        l1 = step.weights_pass(*b1)
        torch.cuda.synchronize()
        assert abs(float(l1) - float(loss1)) <= 2e-5 * abs(float(loss1)), (it, float(l1), float(loss1))
        n = 0
        for (k, p), q in zip(net_a.named_parameters(), net_b.parameters()):
            if any(p is a for a in arch_a) or q.grad is None:
                continue
            close(p.grad, q.grad, ("weights pass", it, k))
            n += 1
        assert n > 1000
        # ---- alpha pass (core/function.py:546-616) ---------------------------------------------------------------------
        sync()
        frozen_before = [p.detach().clone() for p in w_a[:50]]
        im, lpar, lpose, pw = b2
        po, pa = net_b(im)
        losses2 = cq_b(pa, lpar).unsqueeze(0) + cp_b(po, lpose, target_weight=pw).unsqueeze(0)
        if entropy:
            losses2 = losses2 + 2 * net_b.loss_entropy()
        loss2 = 2 * losses2.mean()
        loss2.backward()
        l2 = step.alpha_pass(entropy)(*b2)
        torch.cuda.synchronize()
        assert abs(float(l2) - float(loss2)) <= 2e-5 * abs(float(loss2)), (it, float(l2), float(loss2))
        for i, (a, b) in enumerate(zip(arch_a, arch_b)):
            close(a.grad, b.grad, ("alpha pass", it, i))
        for p, q in zip(w_a[:50], frozen_before):
            assert torch.equal(p.detach(), q)                 # the alpha pass does not touch the weights
    # the passes left every requires_grad flag as they found it
    assert all(p.requires_grad for p in w_a) and all(a.requires_grad for a in arch_a)


def test_search_step_graphed_runs_and_trains():
    """Both passes captured (one hipGraph each): losses finite and falling on a repeated batch, architecture tensors moving."""
    from npp_amd.optim import FusedAdam
    from npp_amd.train_step import SearchStep
    dev = torch.device("cuda:0")
    net, cp, cq, w = _search_setup(dev)
    before = [a.detach().clone() for a in net.arch_parameters()]
    step = SearchStep(net, cp, cq, FusedAdam(w, lr=1e-3), FusedAdam(net.arch_parameters(), lr=3e-3, betas=(0.5, 0.999),
                                                                  weight_decay=0.001), warmup=1)
    b1, b2 = _batch(2, 64, 1, dev), _batch(2, 64, 2, dev)
    losses = [tuple(float(x) for x in step(b1, b2)) for _ in range(5)]
    assert step.weights_pass.graphed and step.alpha_pass(False).graphed
    assert np.isfinite(losses).all() and losses[-1][0] < losses[0][0]
    assert all(float((a - b).abs().max()) > 1e-3 for a, b in zip(net.arch_parameters(), before))


@pytest.mark.parametrize("C,size,batch,max_pix,min_queued,tol", [(32, 96, 4, 9300, 40, 1e-4), (64, 192, 2, 150000, 200, 1e-3)])
def test_batched_weight_gradients_equal_the_immediate_ones(C, size, batch, max_pix, min_queued, tol):
    """TrainStep collects the weight gradients (of maps with <= NPP_DEFER_WGRAD_MAX_PIX pixels) and runs them as one
    npp_conv_wgrad_batched launch per kernel variant before the batched unpack.  Two backward passes over ONE forward (the
    data-gradient chain is deterministic, so both see bit-identical dy): batched == launched where they arise, up to the order of
    the f32 sums -- and the batched launch must actually have been used.  (Round 4: the bound of the C = 64 case was 6e-2 while the SE
    gates summed with float atomics; their slab sums have been deterministic since round 3, so the two passes differ only by the
    order of the weight-gradient kernels' own f32 atomics: 1e-3, and the SE gate weights are compared as well.  A gradient that was
    dropped or written elsewhere is off by 1.0.)"""
    from npp_amd import _ops as K
    from npp_amd.criterion import Criterion_par, Criterion_pose
    from npp_amd.model_augment import Network, set_compute_dtype
    dev = torch.device("cuda:0")
    set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(0)
        net = Network(_cfg(C)).to(dev).train()
        net._auto_graph_off = True
        cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
        im, lpar, lpose, _w = _batch(batch, size, 7, dev)
        output_pose, output_par = net(im)
        loss = (cq(output_par, lpar).unsqueeze(0) + cp(output_pose, lpose).unsqueeze(0)).mean()
        K.DEFER_UNPACK, K.DEFER_WGRAD_MAX_PIX = True, max_pix
        try:
            loss.backward(retain_graph=True)
            queued = len(K._pending_wgrads)
            queued_dw = len(K._pending_dw_wgrads)
            K.flush_wgrads()
            K.flush_unpacks()
        finally:
            K.DEFER_UNPACK, K.DEFER_WGRAD_MAX_PIX = False, 0
        torch.cuda.synchronize()
        batched = {k: p.grad.detach().float().clone() for k, p in net.named_parameters() if p.grad is not None}
        net.zero_grad(set_to_none=True)
        loss.backward()
        torch.cuda.synchronize()
        assert queued >= min_queued and queued_dw >= 20, (queued, queued_dw)
        worst, worst_k = 0.0, None
        checked = 0
        for k, p in net.named_parameters():
            # dense, depthwise and SE-gate conv weights: what the batched launches compute.  (Not compared: a BN bias behind a conv
            # bias -- a pure-noise gradient.)
            if p.grad is None or p.dim() != 4:
                continue
            den = float(p.grad.float().norm())
            if den > 1e-8:
                checked += 1
                e = float((batched[k] - p.grad.float()).norm()) / den
                if e > worst:
                    worst, worst_k = e, k
        assert checked >= min_queued and worst < tol, (checked, worst, worst_k)
    finally:
        set_compute_dtype(torch.float32)
