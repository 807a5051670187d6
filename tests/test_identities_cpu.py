"""Algebraic identities the HIP path relies on where it does not follow the reference's order of operations, checked in f64 on the
CPU with plain torch (no kernel involved):

* MixedOp (model_search_interact.py:39-74): every candidate is `op -> BatchNorm2d(affine=False) [-> Interpolate(bilinear)]` and the
  edge returns the softmax-weighted sum.  npp_amd runs ONE N-sided kernel `sum_k w_k * BN_k(x_k)` on the un-resampled maps and ONE
  Interpolate afterwards (npp_amd/model_search_interact.py: MixedOp.forward, npp_amd/_ops.py: mix_bn_sum).
* its backward takes w_k in gamma's place: d/dx_k and d/dw_k of the fused form equal autograd's through the reference's form.
"""
import pytest
import torch
import torch.nn.functional as F


def _bn(x, eps=1e-5):
    mean = x.mean((0, 2, 3), keepdim=True)
    var = x.var((0, 2, 3), unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


@pytest.mark.parametrize("scale", [2.0, 4.0, 0.5, 0.25, 1.0])
def test_bilinear_resampling_commutes_with_the_mixed_edge(scale):
    torch.manual_seed(0)
    k, n, c, h = 7, 2, 4, 8
    xs = [(torch.randn(n, c, h, h, dtype=torch.float64) * (1 + i) + i).requires_grad_(True) for i in range(k)]
    w = torch.softmax(torch.randn(k, dtype=torch.float64), 0).requires_grad_(True)
    bn_side = [True, True, False, True, True, True, False]      # se_connect / poled_conv end without a BatchNorm

    def f(x, b):
        return _bn(x) if b else x

    def interp(t):
        return F.interpolate(t, scale_factor=scale, mode="bilinear", align_corners=True)

    ref = sum(w[i] * interp(f(xs[i], bn_side[i])) for i in range(k))            # the reference's order
    fused = interp(sum(w[i] * f(xs[i], bn_side[i]) for i in range(k)))          # ours
    assert torch.allclose(ref, fused, rtol=1e-12, atol=1e-12)
    g = torch.randn_like(ref)
    gr = torch.autograd.grad(ref, [w] + xs, g, retain_graph=True)
    gf = torch.autograd.grad(fused, [w] + xs, g)
    for a, b in zip(gr, gf):
        assert torch.allclose(a, b, rtol=1e-10, atol=1e-12)


def test_mixed_edge_backward_is_the_batchnorm_backward_with_w_as_gamma():
    torch.manual_seed(1)
    k, n, c, h = 3, 4, 5, 6
    xs = [(torch.randn(n, c, h, h, dtype=torch.float64) * (1 + i) - i).requires_grad_(True) for i in range(k)]
    w = torch.softmax(torch.randn(k, dtype=torch.float64), 0).requires_grad_(True)
    y = sum(w[i] * _bn(xs[i]) for i in range(k))
    d = torch.randn_like(y)
    gw, *gx = torch.autograd.grad(y, [w] + xs, d)
    count = n * h * h
    s0 = d.sum((0, 2, 3), keepdim=True)
    for i in range(k):
        x = xs[i].detach()
        mean = x.mean((0, 2, 3), keepdim=True)
        inv = 1 / torch.sqrt(x.var((0, 2, 3), unbiased=False, keepdim=True) + 1e-5)
        xh = (x - mean) * inv
        s1 = (d * xh).sum((0, 2, 3), keepdim=True)          # what npp_mix_bn_bwd's reduce accumulates per channel
        dx = w[i].detach() * inv * (d - s0 / count - xh * s1 / count)
        assert torch.allclose(dx, gx[i], rtol=1e-10, atol=1e-12)
        assert torch.allclose(s1.sum(), gw[i], rtol=1e-10, atol=1e-12)


def test_batchnorm_of_batchnorm_closed_form():
    """MixedOp's pooling candidates are `PoolBN -> BatchNorm2d(affine=False)` (model_search_interact.py:48-49 around
    operations.py:44-66): BN2(BN1(p)).  BN1's output has mean 0 and variance v1 = s^2 / (s^2 + eps) per channel exactly, so BN2 needs no
    statistics pass and the pair collapses to ONE operand of the mixed edge (not built yet -- this pins the algebra):
      forward   y = (p - mu) * inv1 * inv2,           inv1 = (s^2 + eps)^-1/2, inv2 = (v1 + eps)^-1/2
      backward  dp = inv1 inv2 (d - mean(d) - kappa * y1 * mean(d y1)),   y1 = (p - mu) inv1,  kappa = 1 + (1 - v1) / (v1 + eps)
      d/dw of w * y = sum_c inv2[c] * sum d y1."""
    torch.manual_seed(2)
    n, c, h, eps = 3, 6, 5, 1e-5
    p = (torch.randn(n, c, h, h, dtype=torch.float64) * torch.linspace(0.01, 3, c, dtype=torch.float64).view(1, c, 1, 1) + 1.5).requires_grad_(True)
    y = _bn(_bn(p, eps), eps)
    d = torch.randn_like(y)
    (gp,) = torch.autograd.grad(y, p, d)
    x = p.detach()
    mu = x.mean((0, 2, 3), keepdim=True)
    s2 = x.var((0, 2, 3), unbiased=False, keepdim=True)
    inv1 = (s2 + eps) ** -0.5
    v1 = s2 * inv1 ** 2
    inv2 = (v1 + eps) ** -0.5
    y1 = (x - mu) * inv1
    assert torch.allclose(y1.mean((0, 2, 3)), torch.zeros(c, dtype=torch.float64), atol=1e-12)
    assert torch.allclose(y1.var((0, 2, 3), unbiased=False), v1.flatten(), rtol=1e-12)
    assert torch.allclose(y.detach(), y1 * inv2, rtol=1e-12, atol=1e-12)
    m0 = d.mean((0, 2, 3), keepdim=True)
    m1 = (d * y1).mean((0, 2, 3), keepdim=True)
    kappa = 1 + (1 - v1) / (v1 + eps)
    dp = inv1 * inv2 * (d - m0 - kappa * y1 * m1)
    assert torch.allclose(dp, gp, rtol=1e-9, atol=1e-12)
    assert torch.allclose((d * y.detach()).sum(), (inv2 * (d * y1).sum((0, 2, 3), keepdim=True)).sum(), rtol=1e-12)


def test_split_rows_and_slices_have_the_gradients_of_plain_indexing():
    """npp_amd._ops.split_rows / split_slices hand out `w[i]` / `w[a:b]` through ONE autograd node (the supernet's cells index the
    softmaxed architecture weights edge by edge, model_search_interact.py:1010-1020 in the reference): same values, same gradients as
    plain indexing, rows nobody used included."""
    from npp_amd import _ops as K
    torch.manual_seed(3)
    w = torch.randn(7, 5, dtype=torch.float64, requires_grad=True)
    rows = K.split_rows(torch.softmax(w, -1))
    assert all(torch.equal(r, torch.softmax(w, -1)[i]) for i, r in enumerate(rows))
    sum((i + 1) * r.pow(2).sum() for i, r in enumerate(rows) if i != 3).backward()      # row 3 is never used
    g1, w.grad = w.grad.clone(), None
    sw = torch.softmax(w, -1)
    sum((i + 1) * sw[i].pow(2).sum() for i in range(7) if i != 3).backward()
    assert torch.allclose(g1, w.grad, rtol=0, atol=1e-15)

    b = torch.randn(14, dtype=torch.float64, requires_grad=True)
    pieces = K.split_slices(b, [3, 4, 5])                                                # the last two entries belong to no piece
    assert [p.shape[0] for p in pieces] == [3, 4, 5]
    sum((k + 2) * torch.softmax(p, -1).pow(2).sum() for k, p in enumerate(pieces)).backward()
    g2, b.grad = b.grad.clone(), None
    (2 * torch.softmax(b[0:3], -1).pow(2).sum() + 3 * torch.softmax(b[3:7], -1).pow(2).sum()
     + 4 * torch.softmax(b[7:12], -1).pow(2).sum()).backward()
    assert torch.allclose(g2, b.grad, rtol=0, atol=1e-15) and float(g2[12:].abs().max()) == 0.0
    # without autograd (eval / no_grad) they are plain views
    with torch.no_grad():
        assert len(K.split_rows(w)) == 7 and K.split_slices(b, [3, 4])[1].shape[0] == 4


@pytest.mark.parametrize("scale", [2, 4, 8])
def test_one_by_one_conv_commutes_with_bilinear_upsampling(scale):
    """model_augment._ResampleConv (the reference's `Interpolate(scale) -> Conv1x1(bias)` cross-task edges, model_augment.py:590-595,
    626-649) runs the conv on the SMALL map and resamples its output when the scale is > 1: a 1x1 conv mixes channels, bilinear
    interpolation mixes pixels with weights that sum to 1, so the two commute -- bias included -- and so do their gradients."""
    torch.manual_seed(scale)
    n, ci, co, h = 2, 6, 4, 5
    x = torch.randn(n, ci, h, h, dtype=torch.float64, requires_grad=True)
    w = torch.randn(co, ci, 1, 1, dtype=torch.float64, requires_grad=True)
    b = torch.randn(co, dtype=torch.float64, requires_grad=True)
    r = torch.randn(n, co, h * scale, h * scale, dtype=torch.float64)

    def run(conv_first):
        for t in (x, w, b):
            t.grad = None
        if conv_first:
            y = F.interpolate(F.conv2d(x, w, b), scale_factor=scale, mode="bilinear", align_corners=True)
        else:
            y = F.conv2d(F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=True), w, b)
        (y * r).sum().backward()
        return y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone()

    ref = run(False)
    swp = run(True)
    for a, c in zip(ref, swp):
        assert torch.allclose(a, c, rtol=1e-12, atol=1e-12)


def test_conv_bias_in_front_of_batch_statistics_batchnorm_has_zero_gradient():
    """_ops._bias_grad: a conv bias that feeds a BatchNorm normalising with BATCH statistics has the exact gradient 0 (the mean
    subtraction removes it); the reference computes a sum that cancels to rounding residue.  The other gradients do not depend on it."""
    torch.manual_seed(1)
    x = torch.randn(3, 5, 6, 6, dtype=torch.float64)
    w = torch.randn(4, 5, 1, 1, dtype=torch.float64, requires_grad=True)
    b = torch.randn(4, dtype=torch.float64, requires_grad=True)
    g = torch.rand(4, dtype=torch.float64, requires_grad=True)
    r = torch.randn(3, 4, 6, 6, dtype=torch.float64)
    y = F.batch_norm(F.conv2d(x, w, b), None, None, g, torch.zeros(4, dtype=torch.float64), True, 0.1, 1e-5)
    (y * r).sum().backward()
    assert float(b.grad.abs().max()) < 1e-12 * float(w.grad.abs().max())
    gw = w.grad.clone()
    w.grad = None
    y0 = F.batch_norm(F.conv2d(x, w, None), None, None, g, torch.zeros(4, dtype=torch.float64), True, 0.1, 1e-5)
    (y0 * r).sum().backward()
    assert torch.allclose(gw, w.grad, rtol=1e-10, atol=1e-12) and torch.allclose(y, y0, rtol=1e-10, atol=1e-12)
