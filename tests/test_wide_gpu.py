"""Merged edges (npp_amd.operations.WideEdges / _ops.conv2d_wide): the ReLU-conv-BN edges of a cell that read the same state run as
ONE conv C -> m C forward and ONE conv m C -> C data gradient (reference: models/genotypes.py:30-54 names the same primitive on the
same state several times; models/model_augment.py:48-62 runs each edge on its own).

Checked against a plain fp32 PyTorch-CPU restatement of `sum_k r_k * BatchNorm_k(conv_k(relu(x)))` (outputs, dx, every weight /
gamma / beta gradient, running statistics) AND against the same modules with merging switched off (K.WIDE = False), on every
kernel family the merged shapes reach: conv_c32 with output / input groups (32 -> 96, 96 -> 32), conv_g4 / conv_h3 (64 k), 1x1."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from tests.helpers import rel_err, rel_l2  # noqa: E402

pytestmark = pytest.mark.gpu


def _make(C, k, m, seed):
    from npp_amd.operations import ReLUConvBN, WideEdges
    g = torch.Generator().manual_seed(seed)
    ops = []
    for i in range(m):
        op = ReLUConvBN(C, C, k, 1, k // 2, affine=True)
        with torch.no_grad():
            op.net[1].weight.copy_(torch.randn(C, C, k, k, generator=g) * (0.3 / (C * k * k) ** 0.5) * 3)
            op.net[2].weight.copy_(torch.rand(C, generator=g) + 0.5)
            op.net[2].bias.copy_(torch.randn(C, generator=g) * 0.2)
        ops.append(op)
    return ops, WideEdges


def _reference(ops, x, rs):
    """fp32 torch-CPU: loss = sum_k (r_k * BN_k(conv_k(relu(x)))).sum(); returns ys, dx, param grads, running stats."""
    x = x.clone().requires_grad_(True)
    ys, params = [], []
    for op, r in zip(ops, rs):
        w = op.net[1].weight.detach().clone().requires_grad_(True)
        ga = op.net[2].weight.detach().clone().requires_grad_(True)
        be = op.net[2].bias.detach().clone().requires_grad_(True)
        rm, rv = torch.zeros_like(ga), torch.ones_like(ga)
        y = F.batch_norm(F.conv2d(F.relu(x), w, None, 1, w.shape[2] // 2), rm, rv, ga, be, True, 0.1, 1e-5)
        ys.append(y)
        params.append((w, ga, be, rm, rv))
    loss = sum((y * r).sum() for y, r in zip(ys, rs))
    loss.backward()
    return [y.detach() for y in ys], x.grad, params


def _run_hip(ops, x_cpu, rs_cpu, dtype, wide, dev):
    from npp_amd import _ops as K
    import copy
    ops = [copy.deepcopy(op).to(dev).train() for op in ops]
    from npp_amd.operations import WideEdges
    if wide:
        WideEdges(ops)
    K.fan_reset()
    K.WIDE = bool(wide)
    try:
        leaf = x_cpu.to(dev).requires_grad_(True)
        x = K.cast(leaf.contiguous(memory_format=torch.channels_last), dtype)
        x = K.bn_add(K.BnSide(x), None)            # (a produced tensor, with a ReLU bit-mask in bf16, as a cell state is)
        before = list(K.WIDE_STATS)
        sides = [op.pending(x) for op in ops]
        ys = [K.bn_add(sd, None, relu=False, training=True) for sd in sides]
        loss = sum((y.float() * r.to(dev)).sum() for y, r in zip(ys, rs_cpu))
        loss.backward()
        torch.cuda.synchronize()
        delta = [a - b for a, b in zip(K.WIDE_STATS, before)]
    finally:
        K.WIDE = True
        K.fan_reset()
    outs = [y.detach().float().cpu() for y in ys]
    grads = [(op.net[1].weight.grad.cpu(), op.net[2].weight.grad.cpu(), op.net[2].bias.grad.cpu(),
              op.net[2].running_mean.cpu(), op.net[2].running_var.cpu()) for op in ops]
    return outs, leaf.grad.float().cpu(), grads, delta


CASES = [
    # C, k, m, H, N
    (32, 3, 3, 96, 6),      # conv_c32: three output groups forward, three input groups in the data gradient (288 tiles)
    (32, 3, 2, 96, 6),      # conv_c32, two groups (DECODER.upsample2's pair at 96^2)
    (32, 3, 3, 20, 2),      # ... and a map conv_c32 does not take (the generic kernel: 96 channels fit no LDS-DMA tile)
    (64, 3, 3, 24, 2),      # conv_g4 64 -> 192 / 192 -> 64
    (128, 3, 2, 24, 2),     # FUSION's pairs (conv_h3 at full size, conv_g4 here)
    (128, 1, 3, 12, 4),     # DECODER.upsample1's three 1x1 on one state
    (64, 1, 2, 24, 2),
    (256, 3, 3, 12, 2),     # ENCODER stage 3
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,k,m,H,N", CASES)
def test_merged_edges_match_torch_and_the_separate_edges(C, k, m, H, N, dtype):
    dev = torch.device("cuda:0")
    ops, _ = _make(C, k, m, seed=C + k + m)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, C, H, H, generator=g)
    rs = [torch.randn(N, C, H, H, generator=g) / (N * H * H) ** 0.5 for _ in range(m)]
    ref_y, ref_dx, ref_p = _reference(ops, x, rs)
    w_y, w_dx, w_p, delta = _run_hip(ops, x, rs, dtype, True, dev)
    s_y, s_dx, s_p, delta_s = _run_hip(ops, x, rs, dtype, False, dev)
    assert delta[0] == 1 and delta[1] == 1 and delta[2] == 0, f"one merged forward, one in-place merged data gradient: {delta}"
    assert delta_s == [0, 0, 0]
    f32 = dtype == torch.float32
    tol_y, tol_g = (2e-4, 5e-4) if f32 else (3e-2, 4e-2)
    err = rel_err if f32 else rel_l2
    for k_ in range(m):
        assert err(w_y[k_], ref_y[k_]) < tol_y, ("y", k_, err(w_y[k_], ref_y[k_]))
        for j, name in enumerate(("dw", "dgamma", "dbeta", "running_mean", "running_var")):
            ref = ref_p[k_][j].grad if j < 3 else ref_p[k_][j]
            e = err(w_p[k_][j], ref)
            assert e < tol_g, (name, k_, e)
            # ... and the merged launch changes nothing but the order of a sum: as close to the separate edges as those are to torch
            assert err(w_p[k_][j], s_p[k_][j]) < tol_g, (name, "vs separate", k_)
        assert err(w_y[k_], s_y[k_]) < (1e-6 if f32 else 1e-2), ("y vs separate", k_)
    assert err(w_dx, ref_dx) < tol_g, ("dx", err(w_dx, ref_dx))
    assert err(w_dx, s_dx) < (1e-5 if f32 else 2e-2), ("dx vs separate", err(w_dx, s_dx))


def test_fixed_genotype_merges_its_duplicated_edges():
    """ENCODER.normal: three std_conv_3x3 on state 0; DECODER.upsample1: three + two std_conv_1x1; FUSION: std_conv_3x3 pairs
    (genotypes.py:30-54) -- the groups the network builds, and the state-dict contract untouched."""
    from types import SimpleNamespace as NS
    from npp_amd.model_augment import Network
    cfg = NS(DATASET=NS(NUM_CLASSES=20, NUM_JOINTS=16), TRAIN=NS(LAYERS=16, INIT_CHANNELS=64),
             MODEL=NS(DECONV_WITH_BIAS=False, HEAD='PSP', REFINE_LAYERS=1))
    net = Network(cfg)
    groups = [g for mod in net.modules() for g in getattr(mod, "_wide_groups", ())]
    normal = [c for c in list(net.cells1) + list(net.cells2) if len(c._wide_groups) == 1 and len(c._wide_groups[0].convs) == 3]
    assert len(normal) == 26                                   # 13 normal cells per branch
    assert sum(len(g.convs) - 1 for g in groups) >= 70          # launches saved per direction
    assert len(net.state_dict()) == 3226                        # names and count as the reference's (tests/test_ddp_cpu.py holds the list)


def _bn_items(n, kind, C, H, N, dtype, dev, seed):
    """n independent fused-add items: kind 'two' = BN(a) + BN(b), 'plain' = BN(a) + b, 'one' = BN(a)."""
    from npp_amd import _ops as K
    g = torch.Generator().manual_seed(seed)
    items = []
    for k in range(n):
        bns = [nn.BatchNorm2d(C).to(dev).train() for _ in range(2)]
        for bn in bns:
            with torch.no_grad():
                bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
                bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        raw = [torch.randn(N, C, H, H, generator=g).to(dev).requires_grad_(True) for _ in range(2)]
        items.append((bns, raw))
    return items


def _run_bn_items(items, kind, dtype, multi, dev, rs):
    from npp_amd import _ops as K
    K.fan_reset()
    K.BN_MULTI = bool(multi)
    try:
        specs = []
        for (bns, raw) in items:
            for r in raw:
                r.grad = None
            for bn in bns:
                bn.weight.grad = bn.bias.grad = None
                bn.running_mean.zero_(); bn.running_var.fill_(1.0)
            xs = [K.cast(r.contiguous(memory_format=torch.channels_last), dtype) for r in raw]
            sa = K.BnSide(xs[0], bns[0], None, private=True)
            sb = None
            if kind == "two":
                sb = K.BnSide(xs[1], bns[1], None, private=True)
            elif kind == "plain":
                sb = K.BnSide(xs[1], private=True)
            specs.append((sa, sb, False, True, None))
        before = list(K.MULTI_STATS)
        ys = K.bn_add_multi(specs)
        loss = sum((y.float() * r).sum() for y, r in zip(ys, rs))
        loss.backward()
        torch.cuda.synchronize()
        delta = [a - b for a, b in zip(K.MULTI_STATS, before)]
    finally:
        K.BN_MULTI = True
        K.fan_reset()
    out = []
    for y, (bns, raw) in zip(ys, items):
        out.append([y.detach().float().cpu()] + [r.grad.float().cpu() if r.grad is not None else None for r in raw] +
                   [t.detach().float().cpu().clone() for bn in bns for t in (bn.weight.grad if bn.weight.grad is not None else torch.zeros(1),
                                                                              bn.bias.grad if bn.bias.grad is not None else torch.zeros(1),
                                                                              bn.running_mean, bn.running_var)])
    return out, delta


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,kind,C,H,N", [(2, "two", 64, 24, 4), (2, "plain", 128, 12, 4), (3, "one", 128, 48, 2), (2, "one", 32, 96, 2),
                                          (4, "two", 256, 6, 2), (3, "plain", 64, 12, 2)])
def test_multi_job_batchnorm_launches_equal_the_single_launches(n, kind, C, H, N, dtype):
    """npp_affine_add_fin_multi / npp_bn_bwd_reduce_multi / npp_bn_bwd_apply_multi run the bodies of the single-job kernels per
    job: outputs, raw-input gradients, gamma / beta gradients and running statistics must agree with n separate launches to the
    last bit of the storage type (the f64 reductions differ only in the order of their atomic adds)."""
    dev = torch.device("cuda:0")
    items = _bn_items(n, kind, C, H, N, dtype, dev, seed=n * 7 + C)
    g = torch.Generator().manual_seed(3)
    rs = [torch.randn(N, C, H, H, generator=g).to(dev) for _ in range(n)]
    m_out, m_delta = _run_bn_items(items, kind, dtype, True, dev, rs)
    s_out, s_delta = _run_bn_items(items, kind, dtype, False, dev, rs)
    assert m_delta == [1, 1, 1, 0], f"one apply, one reduce, one backward-apply launch for all {n} items: {m_delta}"
    assert s_delta == [0, 0, 0, 0]
    tol = 2e-6 if dtype == torch.float32 else 1e-2
    for k in range(n):
        for q, (a, b) in enumerate(zip(m_out[k], s_out[k])):
            if a is None or b is None:
                assert a is None and b is None
                continue
            assert rel_err(a, b) <= tol, (k, q, rel_err(a, b))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,H,N,stride", [(128, 24, 4, 1), (256, 12, 2, 1), (32, 96, 2, 1), (64, 48, 2, 2)])
def test_se_pair_equals_two_single_gates_and_torch(C, H, N, stride, dtype):
    """Two `se_connect` edges on one state (ENCODER.normal, genotypes.py:30-31) as one launch pair (operations.SEPair) against the
    same two modules run one by one, and against a plain fp32 PyTorch-CPU squeeze-excite (operations.py:105-129)."""
    import copy
    from npp_amd import _ops as K
    from npp_amd.operations import SE_Block, SEPair
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C + H)
    mods = [SE_Block(C, stride) for _ in range(2)]
    for m in mods:
        with torch.no_grad():
            for p in (m.conv1.weight, m.conv2.weight):
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
            for p in (m.conv1.bias, m.conv2.bias):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    x_cpu = torch.randn(N, C, H, H, generator=g)
    oh = H // stride
    rs = [torch.randn(N, C, oh, oh, generator=g) for _ in range(2)]

    def run(pair):
        ms = [copy.deepcopy(m).to(dev).train() for m in mods]
        if pair:
            SEPair(ms[0], ms[1])
        K.fan_reset()
        K.SE_PAIR = bool(pair)
        try:
            leaf = x_cpu.to(dev).requires_grad_(True)
            x = K.bn_add(K.BnSide(K.cast(leaf.contiguous(memory_format=torch.channels_last), dtype)), None)
            before = K.SE_PAIR_STATS[0]
            ys = [m(x) for m in ms]
            loss = sum((y.float() * r.to(dev)).sum() for y, r in zip(ys, rs))
            loss.backward()
            torch.cuda.synchronize()
            used = K.SE_PAIR_STATS[0] - before
        finally:
            K.SE_PAIR = True
            K.fan_reset()
        return ([y.detach().float().cpu() for y in ys], leaf.grad.float().cpu(),
                [[p.grad.float().cpu() for p in (m.conv1.weight, m.conv1.bias, m.conv2.weight, m.conv2.bias)] for m in ms], used)

    p_y, p_dx, p_g, used = run(True)
    s_y, s_dx, s_g, used_s = run(False)
    assert used == 1 and used_s == 0
    # torch reference (stride 2: AvgPool2 + BatchNorm follow the gate, operations.py:122-127)
    xr = x_cpu.clone().requires_grad_(True)
    ref_y, ref_params = [], []
    for m in mods:
        ps = [p.detach().clone().requires_grad_(True) for p in (m.conv1.weight, m.conv1.bias, m.conv2.weight, m.conv2.bias)]
        z = F.adaptive_avg_pool2d(xr, 1)
        gate = torch.sigmoid(F.conv2d(F.relu(F.conv2d(z, ps[0], ps[1])), ps[2], ps[3]))
        y = xr * gate
        if stride == 2:
            y = F.batch_norm(F.avg_pool2d(y, 2), None, None, m.bn.weight.detach(), m.bn.bias.detach(), True, 0.1, 1e-5)
        ref_y.append(y)
        ref_params.append(ps)
    sum((y * r).sum() for y, r in zip(ref_y, rs)).backward()
    f32 = dtype == torch.float32
    tol, err = (2e-4, rel_err) if f32 else (3e-2, rel_l2)
    for k in range(2):
        assert err(p_y[k], ref_y[k].detach()) < tol
        assert rel_err(p_y[k], s_y[k]) <= (1e-6 if f32 else 1e-2)
        for j in range(4):
            # (bf16: the gate's parameter gradients are sums of products of rounded pixels that cancel to a fraction of their terms --
            #  the single-gate path shows the same distance to torch; they are held to the single-gate path below)
            if f32:
                assert err(p_g[k][j], ref_params[k][j].grad) < 1e-3, (k, j)
            assert err(p_g[k][j], s_g[k][j]) < (1e-4 if f32 else 3e-2), (k, j, "vs single")
    assert err(p_dx, xr.grad) < (5e-4 if f32 else 3e-2)
    assert err(p_dx, s_dx) < (1e-5 if f32 else 2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layers_on_one_tensor_share_one_data_gradient(dtype):
    """pose_auxlayer (1024 -> 384) and pose_layer (1024 -> 512) both read the concatenated decoder features (model_augment.py:332-351,
    540-548): forward convs of their own, ONE data-gradient conv over the concatenated dy (WideEdges(separate_fwd=True)).  Here at
    256 -> 192 / 128 with biases, against fp32 PyTorch-CPU and against the unmerged modules."""
    import copy
    from npp_amd import _ops as K
    from npp_amd.model_augment import _Layer
    from npp_amd.operations import WideEdges
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    cin, couts, N, H = 256, (192, 128), 2, 24
    layers = []
    for co in couts:
        lay = _Layer(nn.ReLU(), nn.Conv2d(cin, co, 1), nn.BatchNorm2d(co))
        with torch.no_grad():
            lay[1].weight.copy_(torch.randn(co, cin, 1, 1, generator=g) * 0.08)
            lay[1].bias.copy_(torch.randn(co, generator=g) * 0.1)
            lay[2].weight.copy_(torch.rand(co, generator=g) + 0.5)
            lay[2].bias.copy_(torch.randn(co, generator=g) * 0.2)
        layers.append(lay)
    x_cpu = torch.randn(N, cin, H, H, generator=g)
    rs = [torch.randn(N, co, H, H, generator=g) / (N * H * H) ** 0.5 for co in couts]

    def run(wide):
        ls = [copy.deepcopy(m).to(dev).train() for m in layers]
        if wide:
            WideEdges(ls, separate_fwd=True)
        K.fan_reset()
        try:
            leaf = x_cpu.to(dev).requires_grad_(True)
            x = K.bn_add(K.BnSide(K.cast(leaf.contiguous(memory_format=torch.channels_last), dtype)), None)
            before = list(K.WIDE_STATS)
            ys = [m(x) for m in ls]
            sum((y.float() * r.to(dev)).sum() for y, r in zip(ys, rs)).backward()
            torch.cuda.synchronize()
            delta = [a - b for a, b in zip(K.WIDE_STATS, before)]
        finally:
            K.fan_reset()
        return ([y.detach().float().cpu() for y in ys], leaf.grad.float().cpu(),
                [[t.float().cpu() for t in (m[1].weight.grad, m[1].bias.grad, m[2].weight.grad, m[2].bias.grad)] for m in ls], delta)

    w_y, w_dx, w_g, delta = run(True)
    s_y, s_dx, s_g, delta_s = run(False)
    assert delta == [1, 1, 0] and delta_s == [0, 0, 0]
    xr = x_cpu.clone().requires_grad_(True)
    ref_y, ref_p = [], []
    for lay in layers:
        ps = [p.detach().clone().requires_grad_(True) for p in (lay[1].weight, lay[1].bias, lay[2].weight, lay[2].bias)]
        y = F.batch_norm(F.conv2d(F.relu(xr), ps[0], ps[1]), None, None, ps[2], ps[3], True, 0.1, 1e-5)
        ref_y.append(y)
        ref_p.append(ps)
    sum((y * r).sum() for y, r in zip(ref_y, rs)).backward()
    f32 = dtype == torch.float32
    err = rel_err if f32 else rel_l2
    for k in range(2):
        assert err(w_y[k], ref_y[k].detach()) < (2e-4 if f32 else 3e-2)
        assert rel_err(w_y[k], s_y[k]) == 0.0          # the forward launches are the unmerged ones
        for j in range(4):
            if j == 1:
                continue      # (a conv bias in front of a train-mode BatchNorm has a zero gradient: compared to the unmerged path below)
            assert err(w_g[k][j], ref_p[k][j].grad) < (5e-4 if f32 else 4e-2), (k, j)
        for j in range(4):
            assert (w_g[k][j] - s_g[k][j]).abs().max() <= (1e-5 if f32 else 2e-2) * max(float(s_g[k][j].abs().max()), 1e-3), (k, j)
    assert err(w_dx, xr.grad) < (5e-4 if f32 else 4e-2)
    assert err(w_dx, s_dx) < (1e-5 if f32 else 2e-2)


def test_one_by_one_members_of_a_merged_edge_queue_one_weight_gradient_job():
    """Under TrainStep the weight gradients are deferred to one batched launch; the 1x1 members of a merged edge read the same x and
    their dy are channel slices of one buffer, so they queue ONE job with Cout = 256 + 128 (_conv_wgrad_merged) whose rows the batched
    unpack hands to the two parameters.  Against the same modules with one job per member (NPP_WIDE_WGRAD=0 form)."""
    import copy
    from npp_amd import _ops as K
    from npp_amd.model_augment import _Layer
    from npp_amd.operations import WideEdges
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(12)
    cin, couts, N, H = 256, (256, 128), 2, 24
    layers = []
    for co in couts:
        lay = _Layer(nn.ReLU(), nn.Conv2d(cin, co, 1), nn.BatchNorm2d(co))
        with torch.no_grad():
            lay[1].weight.copy_(torch.randn(co, cin, 1, 1, generator=g) * 0.08)
            lay[1].bias.copy_(torch.randn(co, generator=g) * 0.1)
        layers.append(lay)
    x_cpu = torch.randn(N, cin, H, H, generator=g)
    rs = [torch.randn(N, co, H, H, generator=g) / (N * H * H) ** 0.5 for co in couts]

    def run(merged):
        ls = [copy.deepcopy(m).to(dev).train() for m in layers]
        WideEdges(ls, separate_fwd=True)
        old = (K.DEFER_WGRAD_MAX_PIX, K.DEFER_UNPACK, K.WIDE_WGRAD)
        K.fan_reset()
        try:
            K.DEFER_WGRAD_MAX_PIX, K.DEFER_UNPACK, K.WIDE_WGRAD = 1 << 30, True, merged
            leaf = x_cpu.to(dev).requires_grad_(True)
            x = K.bn_add(K.BnSide(K.cast(leaf.contiguous(memory_format=torch.channels_last), torch.bfloat16)), None)
            before = K.WIDE_WGRADS[0]
            ys = [m(x) for m in ls]
            sum((y.float() * r.to(dev)).sum() for y, r in zip(ys, rs)).backward()
            K.flush_wgrads()
            K.flush_unpacks()
            torch.cuda.synchronize()
            jobs = K.WIDE_WGRADS[0] - before
        finally:
            K.DEFER_WGRAD_MAX_PIX, K.DEFER_UNPACK, K.WIDE_WGRAD = old
            K.drop_pending()
            K.fan_reset()
        return [m[1].weight.grad.float().cpu() for m in ls], jobs

    gm, jm = run(True)
    gs, js = run(False)
    assert jm == 1 and js == 0
    for a, b in zip(gm, gs):
        assert a.shape == b.shape and float(b.abs().max()) > 0
        assert (a - b).abs().max() <= 1e-5 * float(b.abs().max())      # (same bf16 operands, f32 accumulation in another order)


@pytest.mark.parametrize("two_sided", [True, False])
@pytest.mark.parametrize("k,c,h", [(3, 128, 96), (3, 32, 96), (1, 128, 24), (3, 64, 24), (1, 128, -96), (1, 256, -96)])
def test_last_writer_delivers_the_batchnorm_backward_sums(two_sided, k, c, h):
    """BN_SUMS: T = BN_a(conv_a(x)) [+ BN_b(conv_b(x))] feeds two ReLU-conv consumers; the data gradient of the consumer that writes
    T's gradient LAST (npp_conv_dgrad_sums on conv_h3 / conv_c32 / conv_g4's 64-row tiles) also delivers sum g, sum g xhat_a, sum g xhat_b,
    and the BatchNorm backward skips its reduce launch.  Against the same graph with NPP_BN_SUMS off: the sums differ only in the order
    of f32 partial sums, so the gradients behind the BatchNorm agree to bf16 rounding."""
    from npp_amd import _ops as K
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(100 + k + c)
    # h < 0: a 1x1 data gradient at N = 16, 96 x 96 -- the persistent shapes whose kernels (conv_g8, conv_g4's 128-row tile) have NO
    # summing epilogue (they used to return NPP_OK with zero sums): the request must be refused, nothing delivered, the
    # BatchNorm backward reduces for itself -- and the gradients still agree
    routed_away = h < 0
    h = abs(h)
    n = 16 if routed_away else (2 if h == 96 else 8)
    x_cpu = torch.randn(n, c, h, h, generator=g)
    wa, wb = (torch.randn(c, c, 1, 1, generator=g) * (1.0 / c ** 0.5) for _ in range(2))
    w1, w2 = (torch.randn(c, c, k, k, generator=g) * (1.0 / (c * k * k) ** 0.5) for _ in range(2))
    r1, r2 = (torch.randn(n, c, h, h, generator=g) for _ in range(2))
    bna, bnb = nn.BatchNorm2d(c).to(dev).train(), nn.BatchNorm2d(c).to(dev).train()

    def run(on):
        old = K.BN_SUMS
        K.BN_SUMS = on
        K.fan_reset()
        try:
            before = list(K.BN_SUMS_STATS)
            leaf = K.cast(x_cpu.to(dev).contiguous(memory_format=torch.channels_last), torch.bfloat16).detach().requires_grad_(True)
            ws = [t.to(dev).requires_grad_(True) for t in (wa, wb, w1, w2)]
            ya, sta = K.conv2d(leaf, ws[0], None, 1, 0, 1, relu_in=False, want_stats=True)
            sides = [K.BnSide(ya, bna, sta)]
            if two_sided:
                yb, stb = K.conv2d(leaf, ws[1], None, 1, 0, 1, relu_in=False, want_stats=True)
                sides.append(K.BnSide(yb, bnb, stb))
            t = K.bn_add(*sides)
            o1, _ = K.conv2d(t, ws[2], None, 1, k // 2, 1, relu_in=True, want_stats=False)
            o2, _ = K.conv2d(t, ws[3], None, 1, k // 2, 1, relu_in=True, want_stats=False)
            gd = lambda r: K.cast(r.to(dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
            torch.autograd.backward([o1, o2], [gd(r1), gd(r2)])
            torch.cuda.synchronize()
            delta = [a - b for a, b in zip(K.BN_SUMS_STATS, before)]
            return leaf.grad.float().cpu(), [w.grad.float().cpu() for w in ws if w.grad is not None], delta
        finally:
            K.BN_SUMS = old
            K.drop_pending()
            K.fan_reset()

    dx1, gw1, d1 = run(True)
    dx0, gw0, d0 = run(False)
    assert d0 == [0, 0, 0]
    if routed_away:
        assert d1[0] >= 1 and d1[1] == 0 and d1[2] == 0, d1      # a ticket was issued, no kernel claimed to have delivered its sums
    else:
        assert d1[0] >= 1 and d1[1] == 1 and d1[2] == 1, d1      # one ticket delivered by the second consumer's data gradient and consumed
    scale = float(dx0.abs().max())
    assert scale > 0 and float((dx1 - dx0).abs().max()) <= 2 ** -6 * scale
    for a, b in zip(gw1, gw0):
        assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max())
