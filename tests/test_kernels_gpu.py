"""Kernel-level GPU parity through the C ABI (via npp_amd._ops) against a plain PyTorch f32 CPU reference of
the same op, on shapes the per-OPS goldens do not reach: every MFMA tile variant (BN = 32/64/128), ragged
M / odd channel counts (3, 6, 16, 20, 2), channel-slice views (ld > C), stride 2, dilation, bias, and
resampling ratios used by the network."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X (no GPU visible)")
    return torch.device("cuda:0")


def _rand(shape, seed):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape).astype(np.float32))


def _to_dev(x_cpu, dtype, slice_pad=0):
    """NHWC device copy; with slice_pad > 0 the tensor is a channel slice of a wider buffer (ld = C + pad)."""
    from npp_amd import _ops as K
    n, c, h, w = x_cpu.shape
    dev = _dev()
    if slice_pad:
        buf = torch.full((n, h, w, c + slice_pad), float("nan"), device=dev).permute(0, 3, 1, 2)
        buf = K.cast(buf, dtype) if dtype != torch.float32 else buf
        view = buf[:, :c]
        tmp = K.cast(x_cpu.to(dev).contiguous(memory_format=torch.channels_last), dtype)
        from npp_amd._lib import lib, check, stream_ptr
        import ctypes as C
        check(lib().npp_copy(C.byref(K.desc(tmp)), C.byref(K.desc(view)), stream_ptr()))
        return view
    return K.cast(x_cpu.to(dev).contiguous(memory_format=torch.channels_last), dtype)


CONV_CASES = [
    # cin, cout, k, stride, pad, dil, H, W, N, relu, bias, slice_pad
    (32, 32, 3, 1, 1, 1, 13, 17, 2, True, False, 0),       # BN=32 tile, ragged M
    (64, 64, 3, 1, 1, 1, 24, 24, 2, True, False, 8),       # BN=64, input is a channel slice
    (128, 128, 3, 1, 1, 1, 24, 24, 1, True, False, 0),     # BN=128
    (128, 256, 1, 1, 0, 1, 12, 12, 3, True, True, 0),      # 1x1 + bias, 2 N-tiles
    (512, 128, 1, 1, 0, 1, 12, 12, 2, True, False, 0),     # long K
    (512, 256, 1, 1, 0, 1, 20, 20, 2, True, False, 0),     # deep 1x1; Cin, Cout % 128 == 0: weight gradient on conv_wgrad_g4
    (64, 128, 3, 2, 1, 1, 20, 20, 2, False, False, 0),     # stem-like stride 2
    (3, 16, 3, 2, 1, 1, 32, 32, 2, False, False, 0),       # image stem: Cin = 3 (padded to 8 in memory)
    (384, 6, 3, 1, 1, 1, 12, 12, 2, True, False, 0),       # edge head: Cout = 6
    (256, 20, 1, 1, 0, 1, 12, 12, 2, False, True, 0),      # parsing head: Cout = 20 + bias
    (128, 16, 1, 1, 0, 1, 12, 12, 2, False, True, 0),      # pose head
    (32, 32, 3, 1, 2, 2, 16, 16, 2, True, False, 0),       # dense dilated (DilConv)
    (32, 32, 3, 1, 1, 1, 40, 40, 2, True, True, 0),        # conv_g4 64x32 tile, two taps per K-tile (Cin = 32), odd tap count, bias
    (128, 32, 1, 1, 0, 1, 24, 24, 2, True, False, 8),      # conv_g4 64x32 tile fwd; dgrad K = 32 (one half-filled K-tile) + mask
    # large maps: bf16 runs on the LDS-DMA kernels (conv_g4.hip 64x64 tiles / conv_g8.hip 256-wide tiles), f32 on conv_s1
    (64, 128, 1, 1, 0, 1, 160, 160, 2, True, True, 0),     # g8 BN=128, 1x1 + bias, one K-tile per output tile
    (128, 256, 1, 1, 0, 1, 225, 225, 1, True, False, 8),   # g8 BN=256, ragged M, channel-slice input; dgrad with mask
    (256, 256, 1, 1, 0, 1, 192, 192, 2, False, False, 0),  # g8 BN=256 fwd + dgrad without mask, 4 K-tiles (deep and wide: not g4)
    (256, 128, 1, 1, 0, 1, 160, 160, 2, False, False, 0),  # g4 on a large map, dgrad N=256
    (32, 32, 3, 1, 1, 1, 50, 96, 10, True, True, 8),       # conv_c32 (12 x 16 tiles, all nine taps resident): ragged last tile row, bias, channel-slice input
    (32, 32, 3, 1, 1, 1, 96, 48, 6, False, False, 0),      # conv_c32 without ReLU
    (64, 128, 3, 1, 1, 1, 225, 225, 1, True, False, 0),    # 3x3 on a large map, ragged M (g8 when NPP_G8_MAXK=3: see below)
    (128, 128, 3, 1, 1, 1, 160, 160, 2, False, True, 0),   # 3x3 + bias, two channel chunks; dgrad without mask
    (128, 256, 1, 1, 0, 1, 200, 201, 2, True, True, 0),    # g4 persistent form: 1258 tiles of 128x128 over 512 slots, ragged M,
                                                           # bias + statistics; dgrad (629 tiles) with the ReLU mask
    # conv_h3.hip: 3x3 with the LDS-resident halo footprint (W % 16 == 0, Cin % 64 == 0, >= 30000 pixels)
    (64, 64, 3, 1, 1, 1, 48, 48, 16, True, False, 0),      # BN=64 tiles, one chunk: the halo is staged once
    (128, 128, 3, 1, 1, 1, 100, 96, 4, True, True, 8),     # ragged tile rows (100 % 8 != 0), channel-slice input, bias; dgrad + mask
    (192, 128, 3, 1, 1, 1, 64, 64, 8, True, False, 0),     # three chunks: the halo double buffer wraps
    (256, 256, 3, 1, 1, 1, 48, 48, 16, False, False, 0),   # two channel tiles, four chunks, no ReLU / mask
    (384, 128, 3, 1, 1, 1, 64, 64, 8, True, False, 0),     # conv_wgrad_h3 (W % 32 == 0): three input-channel tiles, two output tiles
    # conv_thin.hip / conv_wgrad_thin_kernel: the edge head (384 -> 6, model_augment.py:393-398) and its gradients
    (384, 6, 3, 1, 1, 1, 96, 96, 2, True, False, 0),       # the benched shape (N = 2)
    (384, 6, 3, 1, 1, 1, 50, 40, 3, True, False, 8),       # ragged tiles in both directions, channel-slice input
    (128, 2, 3, 1, 1, 1, 24, 24, 2, False, True, 0),       # two output channels, bias, no input ReLU
]
H3_CASES = [c for c in CONV_CASES if c[2] == 3 and c[3] == 1 and c[5] == 1 and c[0] % 64 == 0 and c[7] % 16 == 0
            and c[8] * c[6] * c[7] >= 30000]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(case, dtype, tol):
    from npp_amd import _ops as K
    cin, cout, k, stride, pad, dil, H, W, N, relu, bias, sp = case
    x_cpu = _rand((N, cin, H, W), 1)
    w_cpu = _rand((cout, cin, k, k), 2) * (1.0 / np.sqrt(cin * k * k))
    b_cpu = _rand((cout,), 3) if bias else None
    if dtype == torch.bfloat16:   # compare against the same bf16-rounded operands
        x_cpu = x_cpu.bfloat16().float()
    xr = x_cpu.clone().requires_grad_(True)
    wr = w_cpu.clone().requires_grad_(True)
    wq = wr.bfloat16().float() if dtype == torch.bfloat16 else wr
    br = b_cpu.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(F.relu(xr) if relu else xr, wq, br, stride, pad, dil)
    gy_cpu = _rand(tuple(yr.shape), 4)
    if dtype == torch.bfloat16:
        gy_cpu = gy_cpu.bfloat16().float()
    yr.backward(gy_cpu)

    dev = _dev()
    if cin == 3:
        x = K.image_to_nhwc(x_cpu.to(dev), dtype).detach().requires_grad_(True)
    else:
        x = _to_dev(x_cpu, dtype, sp).detach().requires_grad_(True)
    w = w_cpu.to(dev).requires_grad_(True)
    b = b_cpu.to(dev).requires_grad_(True) if bias else None
    y, st = K.conv2d(x, w, b, stride, pad, dil, relu_in=relu, want_stats=True)
    y.backward(_to_dev(gy_cpu, dtype))
    torch.cuda.synchronize()
    yf = y.detach().float().cpu()
    assert rel_err(yf.numpy(), yr.detach().numpy()) < tol
    # epilogue statistics = statistics of what was stored
    s_ref = torch.cat([yf.double().sum((0, 2, 3)), (yf.double() ** 2).sum((0, 2, 3))])
    assert rel_err(st.view(-1, 2 * cout).sum(0).cpu().numpy(), s_ref.numpy()) < 1e-5
    if cin != 3:
        assert rel_err(x.grad.float().cpu().numpy(), xr.grad.numpy()) < tol
    assert rel_err(w.grad.cpu().numpy(), wr.grad.numpy()) < tol * 2
    if bias:
        assert rel_err(b.grad.cpu().numpy(), br.grad.numpy()) < tol


# The shapes bench.py actually runs on the LDS-DMA kernels (BASELINE config 2: N = 16, 96 x 96 maps; models/model_augment.py:
# 332-351, 371-391): compared with the f32 torch-CPU conv on the same bf16-rounded operands -- forward with input ReLU, bias and the
# statistics epilogue, the data gradient THROUGH THE PRODUCER'S BIT-MASK (NPP_MASK8), the weight gradient both as an immediate
# launch and through the deferred batched launch of TrainStep.
BENCH_CASES = [
    # cin, cout, k, family the fwd + dgrad launches must be counted under, fwd+dgrad launches
    (1024, 512, 1, "conv_g8"),     # pose_layer / par_layer: 16 K-tiles, BN = 256
    (1024, 384, 1, "conv_g8"),     # pose_auxlayer / edge_layer: 384 % 256 != 0 -> the narrower template
    (512, 256, 1, "conv_g8"),      # pose_head / par_head first conv
    (512, 128, 1, "conv_g4"),      # PoseCell1 / ParCell1 preprocess (HBM-bound, persistent 128 x 128 tiles)
    (384, 128, 3, "conv_g4"),      # pose_auxnet (conv_h3, three channel tiles; reports under conv_g4's family)
    (128, 128, 3, "conv_g4"),      # the 30 refine-cell convs (conv_h3)
    (32, 32, 3, "conv_g4"),        # the encoder's first stage (conv_c32: the whole tile problem resident in LDS)
]


@pytest.mark.parametrize("batched_wgrad", [False, True])
@pytest.mark.parametrize("case", BENCH_CASES)
def test_benched_shapes_match_cpu_reference(case, batched_wgrad):
    import ctypes as C
    from npp_amd import _ops as K
    from npp_amd import _lib
    cin, cout, k, fam = case
    N, H, W = 16, 96, 96
    tol = 3e-2
    dev = _dev()
    x_cpu = _rand((N, cin, H, W), 21).bfloat16().float()
    w_cpu = _rand((cout, cin, k, k), 22) * (1.0 / np.sqrt(cin * k * k))
    b_cpu = _rand((cout,), 23)
    xr = x_cpu.clone().requires_grad_(True)
    wr = w_cpu.clone().requires_grad_(True)
    br = b_cpu.clone().requires_grad_(True)
    yr = F.conv2d(F.relu(xr), wr.bfloat16().float(), br, 1, k // 2, 1)
    gy_cpu = _rand(tuple(yr.shape), 24).bfloat16().float()
    yr.backward(gy_cpu)

    x0 = _to_dev(x_cpu, torch.bfloat16).detach().requires_grad_(True)
    x = K.bn_add(K.BnSide(x0))                 # a producer that leaves the ReLU bit-mask of its output (as every cell node does)
    assert K.relu_mask_of(x) is not None
    w = w_cpu.to(dev).requires_grad_(True)
    b = b_cpu.to(dev).requires_grad_(True)
    L = _lib.lib()
    masks_before = K.MASK_STATS[0]
    old = (K.DEFER_WGRAD_MAX_PIX, K.DEFER_UNPACK)
    try:
        if batched_wgrad:
            K.DEFER_WGRAD_MAX_PIX, K.DEFER_UNPACK = 1 << 30, True
        L.npp_prof_begin(_lib.FAM[fam], _lib.NPP_BF16)
        y, st = K.conv2d(x, w, b, 1, k // 2, 1, relu_in=True, want_stats=True)
        y.backward(_to_dev(gy_cpu, torch.bfloat16))
        if batched_wgrad:
            K.flush_wgrads()
            K.flush_unpacks()
        torch.cuda.synchronize()
        ms, fl, by, nl = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        L.npp_prof_end(C.byref(ms), C.byref(fl), C.byref(by), C.byref(nl))
    finally:
        K.DEFER_WGRAD_MAX_PIX, K.DEFER_UNPACK = old
        K.drop_pending()
    assert nl.value == 2, f"{fam}: {nl.value} launches for fwd + dgrad"
    assert K.MASK_STATS[0] == masks_before + 1, "the data gradient did not read the bit-mask"
    yf = y.detach().float().cpu()
    assert rel_err(yf.numpy(), yr.detach().numpy()) < tol
    s_ref = torch.cat([yf.double().sum((0, 2, 3)), (yf.double() ** 2).sum((0, 2, 3))])
    assert rel_err(st.view(-1, 2 * cout).sum(0).cpu().numpy(), s_ref.numpy()) < 1e-5
    assert rel_err(x0.grad.float().cpu().numpy(), xr.grad.numpy()) < tol
    assert rel_err(w.grad.cpu().numpy(), wr.grad.numpy()) < tol
    assert rel_err(b.grad.cpu().numpy(), br.grad.numpy()) < tol


@pytest.mark.parametrize("batched", [False, True, "slabs"])
@pytest.mark.parametrize("cin,cout,k,H,W,N,relu", [
    (128, 128, 3, 12, 12, 3, True),      # three-tap kernel: several image rows per K-tile, P = 432 (ragged last K-tile)
    (256, 128, 3, 13, 17, 2, True),      # odd extents: a border pixel at every position of a fragment half
    (128, 256, 3, 96, 96, 2, False),     # one image row = 1.5 K-tiles
    (128, 128, 3, 6, 40, 5, True),       # H < 64 / W rows per K-tile, image boundaries inside a K-tile
    (384, 128, 3, 24, 24, 4, True),      # three input-channel tiles
    (128, 128, 1, 24, 24, 4, True),      # 1x1 (the 128 x 128 kernel)
    (64, 64, 3, 24, 24, 4, True),        # narrow-layer kernel: four image rows per K-tile
    (32, 32, 3, 96, 96, 2, True),        # narrow-layer kernel: one image row per K-tile (the encoder's first stage)
    (64, 64, 3, 48, 48, 2, False),       # narrow-layer kernel: two rows per K-tile (second stage), no ReLU
    (32, 32, 3, 12, 12, 5, True),        # narrow-layer kernel: eight rows per K-tile, more K-tiles than one workgroup takes
    (32, 32, 3, 6, 128, 3, True),        # narrow-layer kernel: 128-pixel rows
    (64, 64, 3, 13, 17, 2, True),        # narrow channels at an extent the narrow kernel refuses (128 x 128 tile, zero-filled)
    (384, 6, 3, 96, 96, 1, True),        # thin-output kernel (edge head): one image row = 1.5 K-tiles
    (384, 6, 3, 13, 17, 3, True),        # thin-output kernel, odd extents
    (128, 8, 3, 6, 40, 5, False),        # thin-output kernel, 8 output channels, several images per K-tile
    (128, 128, 3, 8, 16, 3, True),       # nine-tap halo kernel (batched): ONE 8 x 16 tile per image, every halo side outside the image
    (64, 128, 3, 16, 32, 2, True),       # halo kernel: 64 input channels (one channel tile), 2 x 2 tiles per image
    (192, 128, 3, 24, 48, 2, False),     # halo kernel: three input-channel tiles, 3 x 3 tiles per image, no ReLU
    (256, 256, 3, 16, 16, 4, True),      # halo kernel: two output- x four input-channel tiles
    (128, 128, 3, 96, 96, 3, True),      # halo kernel: several pixel splits (216 tiles), the benched layer
    (128, 256, 3, 24, 24, 3, False),     # halo kernel on 8 x 8 tiles (maps of whole 8-pixel columns only): 3 x 3 tiles per image, no ReLU
    (128, 128, 3, 8, 8, 5, True),        # halo kernel, 8 x 8 tiles: one tile per image
    (64, 128, 3, 16, 40, 2, True),       # halo kernel, 8 x 8 tiles: 2 x 5 tiles per image, 64 input channels
])
def test_weight_gradient_is_exact_on_integer_data(cin, cout, k, H, W, N, relu, batched):
    """Small-integer activations and gradients: every product and every partial sum is exact in f32 whatever the summation
    order, so the LDS-DMA weight-gradient kernels (conv_wgrad_g4.hip: 128 x 128 tiles, and the three-tap kernel with its
    border masks) must reproduce the f64 reference bit for bit -- a wrong border pixel cannot hide under a bf16 tolerance."""
    from npp_amd import _ops as K
    dev = _dev()
    rng = np.random.default_rng(77)
    x_cpu = torch.from_numpy(rng.integers(-2, 3, (N, cin, H, W)).astype(np.float32))
    gy_cpu = torch.from_numpy(rng.integers(-2, 3, (N, cout, H, W)).astype(np.float32))
    w_cpu = _rand((cout, cin, k, k), 2) * 0.05
    xr = x_cpu.double()
    wr = w_cpu.double().requires_grad_(True)
    F.conv2d(F.relu(xr) if relu else xr, wr, None, 1, k // 2, 1).backward(gy_cpu.double())
    x = _to_dev(x_cpu, torch.bfloat16, 8).detach().requires_grad_(True)        # (a channel slice: ld = C + 8)
    w = w_cpu.to(dev).requires_grad_(True)
    old = (K.DEFER_WGRAD_MAX_PIX, K.DEFER_UNPACK, K.WGRAD_SLABS_BATCHED)
    try:
        if batched:
            K.DEFER_WGRAD_MAX_PIX, K.DEFER_UNPACK = 1 << 30, True
            K.WGRAD_SLABS_BATCHED = batched == "slabs"      # the deterministic form: every pixel split stores a slab of its own
        y, _ = K.conv2d(x, w, None, 1, k // 2, 1, relu_in=relu, want_stats=False)
        y.backward(_to_dev(gy_cpu, torch.bfloat16))
        if batched:
            K.flush_wgrads()
            K.flush_unpacks()
        torch.cuda.synchronize()
    finally:
        K.DEFER_WGRAD_MAX_PIX, K.DEFER_UNPACK, K.WGRAD_SLABS_BATCHED = old
        K.drop_pending()
    got = w.grad.double().cpu()
    assert torch.equal(got, wr.grad), float((got - wr.grad).abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,H,W,N,stride", [(32, 24, 24, 2, 1), (128, 13, 17, 2, 1), (256, 5, 7, 3, 1), (64, 2, 9, 2, 1), (8, 40, 3, 1, 1),
                                             (32, 24, 24, 2, 2), (20, 9, 9, 2, 1)])
def test_max_pool_matches_torch_exactly_with_ties(C, H, W, N, stride, dtype):
    """nn.MaxPool2d(3, stride, 1) (operations.py:55) on heavily quantised data (ties in almost every window): the output AND the
    routing of the gradient (first maximum in (kh, kw) scan order, as ATen) must equal torch's -- the column-walking kernel
    (pool3x3_max_col_kernel) splits the window into row maxima and has to keep that order."""
    from npp_amd import _ops as K
    rng = np.random.default_rng(5)
    x_cpu = torch.from_numpy(rng.integers(-3, 4, (N, C, H, W)).astype(np.float32) * 0.5)
    xr = x_cpu.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, stride, 1)
    gy = torch.from_numpy(rng.integers(-2, 3, tuple(yr.shape)).astype(np.float32))
    yr.backward(gy)
    x = _to_dev(x_cpu, dtype).detach().requires_grad_(True)
    y, st = K.pool3x3(x, False, stride, want_stats=True)
    y.backward(_to_dev(gy, dtype))
    torch.cuda.synchronize()
    assert torch.equal(y.detach().float().cpu(), yr.detach())
    assert torch.equal(x.grad.float().cpu(), xr.grad)
    s_ref = torch.cat([yr.detach().double().sum((0, 2, 3)), (yr.detach().double() ** 2).sum((0, 2, 3))])
    assert rel_err(st.view(-1, 2 * C).sum(0).cpu().numpy(), s_ref.numpy()) < 1e-6


def test_data_gradients_accumulate_in_the_conv_epilogues():
    """One tensor, four conv consumers whose data gradients run on conv_h3 (3x3), conv_g8 (deep 1x1), conv_thin (3x3 -> 6 channels)
    and conv_g4 (1x1): the first stores the shared gradient buffer, the others add into it in their epilogues (NppConvGeom.relu_in
    bit 1) -- no add_n pass -- and the sum equals the f32 reference's (core of VERDICT r2 item 2a; model_augment.py:48-62)."""
    from npp_amd import _ops as K
    dev = _dev()
    N, C, H = 16, 256, 64
    x_cpu = _rand((N, C, H, H), 41).bfloat16().float()
    specs = [(256, 3, True), (256, 1, True), (6, 3, True), (128, 1, False)]
    ws = [_rand((co, C, k, k), 42 + i) * (1.0 / np.sqrt(C * k * k)) for i, (co, k, _r) in enumerate(specs)]
    xr = x_cpu.clone().requires_grad_(True)
    gys = []
    tot = 0
    for (co, k, relu), w in zip(specs, ws):
        yr = F.conv2d(F.relu(xr) if relu else xr, w.bfloat16().float(), None, 1, k // 2)
        gy = _rand(tuple(yr.shape), 50 + co).bfloat16().float()
        gys.append(gy)
        tot = tot + (yr * gy).sum()
    tot.backward()
    x0 = _to_dev(x_cpu, torch.bfloat16).detach().requires_grad_(True)
    x = K.bn_add(K.BnSide(x0))                 # a produced tensor (ReLU bit-mask registered), consumed four times
    before = list(K.FAN_STATS)
    outs = []
    for (co, k, relu), w in zip(specs, ws):
        y, _ = K.conv2d(x, w.to(dev).requires_grad_(True), None, 1, k // 2, 1, relu_in=relu)
        outs.append(y)
    torch.autograd.backward(outs, [_to_dev(g, torch.bfloat16) for g in gys], retain_graph=True)
    torch.cuda.synchronize()
    stored, added, private = (a - b for a, b in zip(K.FAN_STATS, before))
    assert (stored, added, private) == (1, 3, 0), (stored, added, private)
    # a second backward over the same graph starts a fresh buffer (the first one must not be added to again)
    g1 = x0.grad.clone()
    x0.grad = None
    torch.autograd.backward(outs, [_to_dev(g, torch.bfloat16) for g in gys])
    torch.cuda.synchronize()
    assert torch.equal(x0.grad, g1)
    assert rel_err(x0.grad.float().cpu().numpy(), xr.grad.numpy()) < 3e-2


def test_data_gradients_accumulate_in_the_stencil_and_gate_kernels():
    """The same shared gradient buffer through the other cell primitives: a dense conv (stores), then the dilated depthwise conv's
    data gradient, the max-pool backward and the SE gate's backward each ADD into it (NppConvGeom.relu_in bit 1,
    npp_pool3x3_bwd_acc, npp_se_bwd_acc) -- against the f32 reference of the four branches' sum (model_augment.py:48-62)."""
    from npp_amd import _ops as K
    dev = _dev()
    N, C, H = 4, 128, 48
    x_cpu = _rand((N, C, H, H), 61).bfloat16().float()
    wc = _rand((C, C, 3, 3), 62) * (1.0 / np.sqrt(C * 9))
    wd = _rand((C, 1, 3, 3), 63) * 0.3
    w1, b1 = _rand((C // 2, C, 1, 1), 64) * (1.0 / np.sqrt(C)), _rand((C // 2,), 65) * 0.1
    w2, b2 = _rand((C, C // 2, 1, 1), 66) * (1.0 / np.sqrt(C // 2)), _rand((C,), 67) * 0.1
    xr = x_cpu.clone().requires_grad_(True)
    ys = [F.conv2d(F.relu(xr), wc.bfloat16().float(), None, 1, 1),
          F.conv2d(F.relu(xr), wd, None, 1, 2, 2, groups=C),
          F.max_pool2d(xr, 3, 1, 1),
          xr * torch.sigmoid(F.conv2d(F.relu(F.conv2d(xr.mean((2, 3), keepdim=True), w1, b1)), w2, b2))]
    gys = [_rand(tuple(y.shape), 70 + i).bfloat16().float() for i, y in enumerate(ys)]
    sum((y * g).sum() for y, g in zip(ys, gys)).backward()
    x0 = _to_dev(x_cpu, torch.bfloat16).detach().requires_grad_(True)
    x = K.bn_add(K.BnSide(x0))
    before = list(K.FAN_STATS)
    P = lambda t: t.to(dev).requires_grad_(True)      # noqa: E731
    outs = [K.conv2d(x, P(wc), None, 1, 1, 1, relu_in=True)[0],
            K.dwconv2d(x, P(wd), 1, 2, 2, relu_in=True),
            K.pool3x3(x, False, 1)[0],
            K.se_scale(x, P(w1), P(b1), P(w2), P(b2))]
    for o, r in zip(outs, ys):
        assert rel_err(o.detach().float().cpu().numpy(), r.detach().numpy()) < 3e-2
    torch.autograd.backward(outs, [_to_dev(g, torch.bfloat16) for g in gys])
    torch.cuda.synchronize()
    stored, added, private = (a - b for a, b in zip(K.FAN_STATS, before))
    assert (stored, added, private) == (1, 3, 0), (stored, added, private)
    assert rel_err(x0.grad.float().cpu().numpy(), xr.grad.numpy()) < 3e-2


def _g8_launch_count(cin, cout, k, fam="conv_g8", hw=192):
    """fwd + dgrad of one conv in bf16 -> number of launches of the given conv kernel family (profiler family counter)."""
    import ctypes as C
    from npp_amd import _ops as K
    from npp_amd import _lib
    L = _lib.lib()
    dev = _dev()
    x = K.cast(torch.randn(2, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
    x.requires_grad_(True)
    w = (torch.randn(cout, cin, k, k, device=dev) * 0.03).requires_grad_(True)
    L.npp_prof_begin(_lib.FAM[fam], _lib.NPP_BF16)
    y, _ = K.conv2d(x, w, None, 1, k // 2, 1, relu_in=True, want_stats=True)
    y.backward(K.cast(torch.randn(2, cout, hw, hw, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16))
    torch.cuda.synchronize()
    ms, fl, by, nl = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
    L.npp_prof_end(C.byref(ms), C.byref(fl), C.byref(by), C.byref(nl))
    return nl.value


def test_shapes_take_their_lds_dma_kernels():
    """Deep and wide 1x1 convs on large maps run on conv_g8_kernel, small maps / 64-channel layers / the 128->128 3x3 on
    conv_g4_kernel (fwd and dgrad), in bf16 -- not on a fallback."""
    assert _g8_launch_count(256, 256, 1, "conv_g8", 192) == 2
    assert _g8_launch_count(128, 128, 3, "conv_g4", 24) == 2
    assert _g8_launch_count(64, 64, 3, "conv_g4", 48) == 2
    assert _g8_launch_count(128, 128, 3, "conv_g4", 160) == 2        # (conv_h3 reports under the same family)
    assert _g8_launch_count(32, 32, 3, "conv_g4", 96) == 2
    assert _g8_launch_count(128, 32, 1, "conv_g4", 96) == 2
    assert _g8_launch_count(384, 6, 3, "conv_g4", 96) == 2           # the edge head: conv_thin.hip, same family


# ---- cases that need a process of their own (switches the library reads once per process): background children, tests/bg_children.py
import os      # noqa: E402
import sys as _sys      # noqa: E402
import tempfile as _tempfile      # noqa: E402
import bg_children      # noqa: E402

_HERE = os.path.dirname(os.path.abspath(__file__))
for _cfg in ("1", "3", "4"):
    bg_children.register(f"h3-cfg-{_cfg}", [_sys.executable, os.path.join(_HERE, "h3_cfg_worker.py")], dict(NPP_H3_CFG=_cfg), timeout=600)
_EPI_DIR = _tempfile.mkdtemp(prefix="npp_epi_")
for _lean in ("0", "1"):
    bg_children.register(f"epi-{_lean}", [_sys.executable, os.path.join(_HERE, "epi_worker.py"), os.path.join(_EPI_DIR, f"epi{_lean}.npz")],
                         dict(NPP_EPI_LEAN=_lean, NPP_EPI_CENSUS="1"), timeout=600)
bg_children.register("wgrad-slabs", [_sys.executable, os.path.join(_HERE, "wgrad_slabs_worker.py")], dict(NPP_WGRAD_SLABS="1"), timeout=600)
for _sk in ("2", "3"):
    bg_children.register(f"g4-splitk-{_sk}", [_sys.executable, os.path.join(_HERE, "g4_splitk_worker.py")],
                         dict(NPP_G4_SPLITK=_sk, NPP_G4_SPLIT_DBG="1"), timeout=600)
bg_children.register("g8-taps", [_sys.executable, os.path.join(_HERE, "g8_taps_worker.py")],
                     dict(NPP_G8_MAXK="3", NPP_DISABLE_G4="1", NPP_DISABLE_H3="1"), timeout=600)


@pytest.mark.parametrize("cfg", ["1", "3", "4"])
def test_h3_tile_configurations_in_subprocess(cfg):
    """conv_h3's other tile shapes (NPP_H3_CFG, read once per process): 16-row / 12-row / 8-row tiles with 8 waves -- the
    halo-footprint parity cases through each of them, and the launches really are conv_h3's."""
    r = bg_children.result(f"h3-cfg-{cfg}")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "h3 cfg ok" in r.stdout


def test_lean_epilogues_equal_the_generic_ones():
    """conv_epi.h / conv_g8's epilogue_lean against the kernels' generic epilogues (NPP_EPI_LEAN=0, read once per process): the stored
    bf16 outputs -- forward, data gradient of a first writer through the bit-mask, accumulating second writer -- must be BIT-identical
    (same values, same rounding, another instruction sequence); the BatchNorm statistics are summed in another order (reduce-scatter
    instead of all-reduce over the 16 pixel lanes): equal to 1e-6 of the largest entry."""
    res = {}
    for lean in ("0", "1"):
        r = bg_children.result(f"epi-{lean}")
        assert r.returncode == 0 and "epi ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        res[lean] = np.load(os.path.join(_EPI_DIR, f"epi{lean}.npz"))
        # the cases reach every kernel that has a lean epilogue, in the three formats the network runs
        census = [ln for ln in r.stderr.splitlines() if ln.startswith("npp-epi ")]
        for kern in ("conv_g8", "conv_g4", "conv_h3", "conv_c32"):
            for fmt in ("stats1 mask0 accum0", "stats0 mask1 accum0", "stats0 mask1 accum1"):
                assert any(ln.split()[1] == kern and fmt in ln for ln in census), (kern, fmt)
    a, b = res["0"], res["1"]
    assert set(a.files) == set(b.files) and len(a.files) >= 35
    for k in a.files:
        if "_st" in k:
            assert np.abs(a[k] - b[k]).max() <= 1e-6 * np.abs(a[k]).max(), k
        else:
            assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k
        assert np.abs(a[k].astype(np.float64)).max() > 0, k


def test_deterministic_slab_weight_gradient_in_subprocess():
    """NPP_WGRAD_SLABS=1 (read once per process): the wide-map 3x3 weight gradients through conv_wgrad_h3's split-K slabs +
    npp_unpack_wgrad_sum -- parity cases, and two runs give bit-identical gradients (no float atomics on that path)."""
    r = bg_children.result("wgrad-slabs")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "wgrad slabs ok" in r.stdout


@pytest.mark.parametrize("shares", ["2", "3"])
def test_g4_split_k_in_subprocess(shares):
    """conv_g4's split-K form for grids of fewer tiles than CUs (NPP_G4_SPLITK forces the number of shares, read once per process):
    parity of forward / data gradient / weight gradient on five shapes, bit-identical repeats, and the launches really were split."""
    r = bg_children.result(f"g4-splitk-{shares}")
    assert r.returncode == 0 and "g4 splitk ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stderr.splitlines() if ln.startswith("npp-g4-split")]
    assert sum(1 for ln in lines if ln.endswith(f"-> S {shares} (wanted {shares})")) >= 8, lines[:20]


def test_g8_taps_variant_in_subprocess():
    """The KxK (per-tap shift, zero border by out-of-range DMA) variant of conv_g8_kernel is opt-in (NPP_G8_MAXK=3, read once
    per process): run the large-map 3x3 parity cases on it in ONE child process."""
    r = bg_children.result("g8-taps")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "g8 taps ok" in r.stdout


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("C,k,stride,dil,H", [(32, 3, 1, 2, 24), (128, 3, 2, 4, 24), (64, 5, 1, 2, 16), (1024, 3, 1, 2, 6),
                                                 (256, 3, 2, 2, 12)])
def test_dwconv_fwd_bwd(C, k, stride, dil, H, dtype, tol):
    from npp_amd import _ops as K
    pad = dil * (k - 1) // 2
    N = 2
    x_cpu = _rand((N, C, H, H), 5)
    w_cpu = _rand((C, 1, k, k), 6) * 0.3
    if dtype == torch.bfloat16:
        x_cpu = x_cpu.bfloat16().float()
    xr, wr = x_cpu.clone().requires_grad_(True), w_cpu.clone().requires_grad_(True)
    yr = F.conv2d(F.relu(xr), wr, None, stride, pad, dil, groups=C)
    gy = _rand(tuple(yr.shape), 7)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().float()
    yr.backward(gy)
    x = _to_dev(x_cpu, dtype).detach().requires_grad_(True)
    w = w_cpu.to(_dev()).requires_grad_(True)
    y = K.dwconv2d(x, w, stride, pad, dil, relu_in=True)
    y.backward(_to_dev(gy, dtype))
    torch.cuda.synchronize()
    assert rel_err(y.detach().float().cpu().numpy(), yr.detach().numpy()) < tol
    assert rel_err(x.grad.float().cpu().numpy(), xr.grad.numpy()) < tol
    assert rel_err(w.grad.cpu().numpy(), wr.grad.numpy()) < tol * 2


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("C,hin,hout", [(32, 12, 24), (256, 12, 96), (64, 24, 12), (128, 24, 6), (16, 13, 29), (512, 6, 48), (20, 24, 96), (6, 7, 30)])
def test_bilinear_fwd_bwd(C, hin, hout, dtype, tol):
    from npp_amd import _ops as K
    x_cpu = _rand((2, C, hin, hin), 8)
    if dtype == torch.bfloat16:
        x_cpu = x_cpu.bfloat16().float()
    xr = x_cpu.clone().requires_grad_(True)
    yr = F.interpolate(xr, size=(hout, hout), mode='bilinear', align_corners=True)
    gy = _rand(tuple(yr.shape), 9)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().float()
    yr.backward(gy)
    x = _to_dev(x_cpu, dtype).detach().requires_grad_(True)
    y = K.bilinear(x, hout, hout)
    y.backward(_to_dev(gy, dtype))
    torch.cuda.synchronize()
    assert rel_err(y.detach().float().cpu().numpy(), yr.detach().numpy()) < tol
    assert rel_err(x.grad.float().cpu().numpy(), xr.grad.numpy()) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("C,H,W,N", [(32, 24, 24, 2), (64, 13, 17, 3), (256, 12, 12, 16), (128, 96, 96, 2), (512, 6, 6, 2), (20, 9, 9, 2)])
def test_se_gate_matches_torch_and_is_reproducible(C, H, W, N, dtype, tol):
    """SE_Block's gate (operations.py:118-123) on the two-launch kernels of se.hip vs plain torch: output, dx and the four
    parameter gradients; two runs give bit-identical results (slab sums in a fixed order, no float atomics)."""
    from npp_amd import _ops as K
    dev = _dev()
    x_cpu = _rand((N, C, H, W), 31) + 0.3
    if dtype == torch.bfloat16:
        x_cpu = x_cpu.bfloat16().float()
    w1 = _rand((C // 2, C, 1, 1), 32) * (1.0 / np.sqrt(C))
    b1 = _rand((C // 2,), 33) * 0.1
    w2 = _rand((C, C // 2, 1, 1), 34) * (1.0 / np.sqrt(C // 2))
    b2 = _rand((C,), 35) * 0.1
    gy = _rand((N, C, H, W), 36)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().float()
    ref = [t.clone().requires_grad_(True) for t in (x_cpu, w1, b1, w2, b2)]
    g = torch.sigmoid(F.conv2d(F.relu(F.conv2d(ref[0].mean((2, 3), keepdim=True), ref[1], ref[2])), ref[3], ref[4]))
    (ref[0] * g).backward(gy)
    yr = (ref[0] * g).detach()
    runs = []
    for _ in range(2):
        x = _to_dev(x_cpu, dtype).detach().requires_grad_(True)
        ps = [t.to(dev).requires_grad_(True) for t in (w1, b1, w2, b2)]
        y = K.se_scale(x, *ps)
        y.backward(_to_dev(gy, dtype))
        torch.cuda.synchronize()
        runs.append([y.detach().float().cpu(), x.grad.float().cpu()] + [p.grad.cpu() for p in ps])
    y, dx, dw1, db1, dw2, db2 = runs[0]
    assert rel_err(y.numpy(), yr.numpy()) < tol
    assert rel_err(dx.numpy(), ref[0].grad.numpy()) < tol
    ptol = tol if dtype == torch.float32 else 5e-2
    for got, r in zip((dw1, db1, dw2, db2), ref[1:]):
        assert rel_err(got.numpy(), r.grad.numpy()) < ptol * 5
    if C % (8 if dtype == torch.bfloat16 else 4) == 0 and K.SE_FUSED:
        for a, b in zip(runs[0], runs[1]):
            assert torch.equal(a, b), "SE sums are not reproducible"


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("C,H,N,relu,two", [(32, 24, 2, False, True), (128, 12, 4, True, False), (6, 12, 2, True, False),
                                              (384, 6, 2, False, True), (1024, 3, 2, False, False)])
def test_batchnorm_fused_add(C, H, N, relu, two, dtype, tol):
    """out = relu?(BN(a) [+ BN(b)]) in train mode vs nn.BatchNorm2d, incl. running stats and all grads."""
    from npp_amd import _ops as K
    dev = _dev()
    a_cpu, b_cpu = _rand((N, C, H, H), 10) * 2 + 0.5, _rand((N, C, H, H), 11)
    if dtype == torch.bfloat16:
        a_cpu, b_cpu = a_cpu.bfloat16().float(), b_cpu.bfloat16().float()
    bns = [torch.nn.BatchNorm2d(C, momentum=0.1) for _ in range(2)]
    for i, bn in enumerate(bns):
        with torch.no_grad():
            bn.weight.copy_(_rand((C,), 12 + i) * 0.2 + 1)
            bn.bias.copy_(_rand((C,), 14 + i) * 0.1)
    import copy
    ref = copy.deepcopy(bns)
    ar, br = a_cpu.clone().requires_grad_(True), b_cpu.clone().requires_grad_(True)
    yr = ref[0](ar) + (ref[1](br) if two else 0)
    if relu:
        yr = F.relu(yr)
    gy = _rand(tuple(yr.shape), 16)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().float()
    yr.backward(gy)
    for bn in bns:
        bn.to(dev)
    a = _to_dev(a_cpu, dtype).detach().requires_grad_(True)
    b = _to_dev(b_cpu, dtype).detach().requires_grad_(True)
    y = K.bn_add(K.BnSide(a, bns[0]), K.BnSide(b, bns[1]) if two else None, relu=relu, training=True)
    y.backward(_to_dev(gy, dtype))
    torch.cuda.synchronize()
    assert rel_err(y.detach().float().cpu().numpy(), yr.detach().numpy()) < tol
    assert rel_err(a.grad.float().cpu().numpy(), ar.grad.numpy()) < tol * 5
    assert rel_err(bns[0].weight.grad.cpu().numpy(), ref[0].weight.grad.numpy()) < tol * 5
    assert rel_err(bns[0].bias.grad.cpu().numpy(), ref[0].bias.grad.numpy()) < tol * 5
    assert rel_err(bns[0].running_mean.cpu().numpy(), ref[0].running_mean.numpy()) < tol
    assert rel_err(bns[0].running_var.cpu().numpy(), ref[0].running_var.numpy()) < tol
    assert int(bns[0].num_batches_tracked) == 1
    if two:
        assert rel_err(b.grad.float().cpu().numpy(), br.grad.numpy()) < tol * 5
        assert rel_err(bns[1].weight.grad.cpu().numpy(), ref[1].weight.grad.numpy()) < tol * 5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1e-4)])
@pytest.mark.parametrize("C,H,N,relu,mode", [(64, 24, 4, True, "two"), (128, 12, 2, False, "one"), (256, 6, 2, True, "plain_b"),
                                               (512, 24, 2, True, "two"), (32, 48, 2, False, "plain_b")])
def test_bn_fused_prologues_equal_the_separate_kernels(C, H, N, relu, mode, dtype, tol):
    """npp_affine_add_fin / npp_bn_bwd_reduce(2)_acc / npp_bn_bwd_apply(2)_fin (finalize and coefficient arithmetic in the prologue of
    the elementwise kernels) against npp_bn_finalize(2) + npp_affine_add and npp_bn_bwd_reduce(2) + coeffs(2) + apply(2): same
    outputs (bit-equal: same arithmetic per element), same running statistics, same gradients up to the f64 summation order."""
    from npp_amd import _ops as K
    import copy
    dev = _dev()
    a_cpu, b_cpu = _rand((N, C, H, H), 20) * 2 + 0.5, _rand((N, C, H, H), 21)
    gy = _rand((N, C, H, H), 22)
    bns0 = [torch.nn.BatchNorm2d(C, momentum=0.1) for _ in range(2)]
    for i, bn in enumerate(bns0):
        with torch.no_grad():
            bn.weight.copy_(_rand((C,), 23 + i) * 0.2 + 1)
            bn.bias.copy_(_rand((C,), 25 + i) * 0.1)

    def run(fused):
        K.FUSE_BN_FIN = fused
        bns = [copy.deepcopy(bn).to(dev) for bn in bns0]
        a = _to_dev(a_cpu, dtype).detach().requires_grad_(True)
        b = _to_dev(b_cpu, dtype).detach().requires_grad_(True)
        sb = None if mode == "one" else (K.BnSide(b, bns[1]) if mode == "two" else K.BnSide(b))
        y = K.bn_add(K.BnSide(a, bns[0]), sb, relu=relu, training=True)
        y.backward(_to_dev(gy, dtype))
        torch.cuda.synchronize()
        res = [y.detach().float().cpu().numpy(), a.grad.float().cpu().numpy(), bns[0].weight.grad.cpu().numpy(),
               bns[0].bias.grad.cpu().numpy(), bns[0].running_mean.cpu().numpy(), bns[0].running_var.cpu().numpy()]
        assert int(bns[0].num_batches_tracked) == 1
        if mode != "one":
            res.append(b.grad.float().cpu().numpy())
        if mode == "two":
            res += [bns[1].weight.grad.cpu().numpy(), bns[1].running_var.cpu().numpy()]
            assert int(bns[1].num_batches_tracked) == 1
        return res

    try:
        fused, plain = run(True), run(False)
    finally:
        K.FUSE_BN_FIN = True
    for i, (f, q) in enumerate(zip(fused, plain)):
        assert rel_err(f, q) < tol, (i, rel_err(f, q))


@pytest.mark.parametrize("C,H,W,N,mode,pad", [
    (32, 96, 96, 16, "two", 0),      # 9 items per thread, 256 blocks (the encoder's first stage at the bench's batch)
    (32, 96, 96, 16, "one", 0),
    (64, 48, 48, 16, "two", 8),      # operands that are channel slices (ld = C + 8)
    (128, 24, 24, 16, "two", 0),     # 5 items per thread
    (256, 12, 12, 16, "one", 0),
    (512, 12, 12, 2, "two", 0),      # 64 column vectors: one wave per pixel row
    (8, 5, 7, 3, "two", 0),          # one column vector, a handful of items: one block
    (16, 13, 17, 2, "one", 8),
])
def test_batchnorm_backward_in_one_launch_equals_the_two_launch_form(C, H, W, N, mode, pad):
    """npp_bn_bwd_one / npp_bn_bwd_one2 (csrc/bn_one.hip: reduce + grid barrier + apply, tensors held in registers) against
    npp_bn_bwd_reduce(2)_acc + npp_bn_bwd_apply(2)_fin on the same inputs: dx bit-equal up to the f64 summation order of the sums
    (bf16 outputs: equal or one ulp apart), dgamma / dbeta to 1e-5; and against the f64 torch BatchNorm backward.  Every call is
    repeated so that the barrier counter is used by several launches in a row."""
    from npp_amd import _ops as K
    import copy
    dev = _dev()
    dtype = torch.bfloat16
    a_cpu, b_cpu = (_rand((N, C, H, W), 20) * 2 + 0.5).bfloat16().float(), _rand((N, C, H, W), 21).bfloat16().float()
    gy = _rand((N, C, H, W), 22).bfloat16().float()
    bns0 = [torch.nn.BatchNorm2d(C, momentum=0.1) for _ in range(2)]
    for i, bn in enumerate(bns0):
        with torch.no_grad():
            bn.weight.copy_(_rand((C,), 23 + i) * 0.2 + 1)
            bn.bias.copy_(_rand((C,), 25 + i) * 0.1)

    def run(one):
        K.BN_ONE = one
        res = None
        for _rep in range(3):
            bns = [copy.deepcopy(bn).to(dev) for bn in bns0]
            a = _to_dev(a_cpu, dtype, pad).detach().requires_grad_(True)
            b = _to_dev(b_cpu, dtype, pad).detach().requires_grad_(True)
            sb = K.BnSide(b, bns[1]) if mode == "two" else None
            y = K.bn_add(K.BnSide(a, bns[0]), sb, relu=False, training=True)
            y.backward(_to_dev(gy, dtype))
            torch.cuda.synchronize()
            cur = [a.grad.float().cpu(), bns[0].weight.grad.cpu(), bns[0].bias.grad.cpu()]
            if mode == "two":
                cur += [b.grad.float().cpu(), bns[1].weight.grad.cpu(), bns[1].bias.grad.cpu()]
            if res is not None:
                for q, r in zip(cur, res):
                    assert rel_err(q.numpy(), r.numpy()) < 1e-5
            res = cur
        return res

    before = list(K.BN_ONE_STATS)
    try:
        one, two = run(True), run(False)
    finally:
        K.BN_ONE = False
    k = 1 if mode == "two" else 0
    assert K.BN_ONE_STATS[k] == before[k] + 3 and K.BN_ONE_STATS[1 - k] == before[1 - k], "the one-launch kernel did not run"
    for i, (f, q) in enumerate(zip(one, two)):
        big = i in (0, 3)
        assert rel_err(f.numpy(), q.numpy()) < (2e-3 if big else 1e-5), (i, rel_err(f.numpy(), q.numpy()))
        if big:      # bf16 outputs of the same f32 arithmetic: almost all elements identical
            assert float((f != q).float().mean()) < 0.02
    # f64 reference
    ar, br = a_cpu.double().requires_grad_(True), b_cpu.double().requires_grad_(True)
    ref = [copy.deepcopy(bn).double() for bn in bns0]
    yr = ref[0](ar) + (ref[1](br) if mode == "two" else 0)
    yr.backward(gy.double())
    assert rel_err(one[0].numpy(), ar.grad.numpy()) < 1e-2
    assert rel_err(one[1].numpy(), ref[0].weight.grad.numpy()) < 2e-3
    if mode == "two":
        assert rel_err(one[3].numpy(), br.grad.numpy()) < 1e-2


def test_pair_of_fused_batchnorm_adds_equals_two_single_ones():
    """K.bn_add_pair (_ops._BnAddPair: two fused BatchNorm adds as ONE autograd node -- under SyncBatchNorm their backward passes
    share one statistics exchange) against two K.bn_add calls on the same inputs, local statistics: outputs bit-equal, every
    gradient equal up to the order of the f64 atomics; one operand of the second add is a plain tensor."""
    from npp_amd import _ops as K
    import copy
    dev = _dev()
    C, H, N = 64, 24, 4
    xs = [(_rand((N, C, H, H), 80 + i) * (1 + 0.2 * i)).bfloat16().float() for i in range(4)]
    gs = [_rand((N, C, H, H), 90 + i).bfloat16().float() for i in range(2)]
    bns0 = [torch.nn.BatchNorm2d(C) for _ in range(3)]
    for i, bn in enumerate(bns0):
        with torch.no_grad():
            bn.weight.copy_(_rand((C,), 95 + i) * 0.2 + 1)
            bn.bias.copy_(_rand((C,), 98 + i) * 0.1)

    def run(pair):
        bns = [copy.deepcopy(bn).to(dev) for bn in bns0]
        t = [_to_dev(x, torch.bfloat16).detach().requires_grad_(True) for x in xs]
        s0 = (K.BnSide(t[0], bns[0]), K.BnSide(t[1], bns[1]), False, True, None)
        s1 = (K.BnSide(t[2], bns[2]), K.BnSide(t[3]), False, True, None)      # BN(x2) + x3 (plain)
        if pair:
            y0, y1 = K.bn_add_pair(s0, s1)
        else:
            y0, y1 = K.bn_add(*s0), K.bn_add(*s1)
        torch.autograd.backward([y0, y1], [_to_dev(gs[0], torch.bfloat16), _to_dev(gs[1], torch.bfloat16)])
        torch.cuda.synchronize()
        return ([y0.detach().float().cpu(), y1.detach().float().cpu()] + [x.grad.float().cpu() for x in t]
                + [p.grad.cpu() for bn in bns for p in (bn.weight, bn.bias)] + [bn.running_var.cpu() for bn in bns])

    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for i, (u, v) in enumerate(zip(a[2:], b[2:])):
        assert rel_err(u.numpy(), v.numpy()) < 1e-4, i


def test_batchnorm_one_launch_kernels_of_two_streams_do_not_wait_on_each_other():
    """Two streams run the grid-barrier kernel at the same time, 200 launches each (every stream has barrier counters of its own):
    all blocks of both launches must be resident together (csrc/bn_one.hip's register / grid-size bound) -- a deadlock would hang
    here -- and every launch gives the result of the first."""
    from npp_amd import _ops as K
    dev = _dev()
    C, H, N = 32, 96, 16
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    torch.cuda.synchronize()
    for si, st in enumerate(streams):
        with torch.cuda.stream(st):
            bna, bnb = torch.nn.BatchNorm2d(C).to(dev), torch.nn.BatchNorm2d(C).to(dev)
            a = _to_dev(_rand((N, C, H, H), 30 + si), torch.bfloat16).detach().requires_grad_(True)
            b = _to_dev(_rand((N, C, H, H), 32 + si), torch.bfloat16).detach().requires_grad_(True)
            g = _to_dev(_rand((N, C, H, H), 34 + si), torch.bfloat16)
            outs.append((a, b, g, bna, bnb, []))
    torch.cuda.synchronize()
    before = K.BN_ONE_STATS[1]
    K.BN_ONE = True
    try:
        for it in range(200):
            for si, st in enumerate(streams):
                a, b, g, bna, bnb, got = outs[si]
                with torch.cuda.stream(st):
                    a.grad = None
                    y = K.bn_add(K.BnSide(a, bna), K.BnSide(b, bnb), relu=False, training=True)
                    y.backward(g)
                    if it in (0, 199):
                        got.append(a.grad.clone())
        torch.cuda.synchronize()
    finally:
        K.BN_ONE = False
    assert K.BN_ONE_STATS[1] == before + 400
    for a, b, g, bna, bnb, got in outs:
        assert rel_err(got[1].float().cpu().numpy(), got[0].float().cpu().numpy()) < 1e-3


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("C,H,N,pattern", [(16, 24, 2, "BBPBBBP"), (32, 12, 4, "BPB"), (64, 6, 2, "BBBBBBBB"), (8, 12, 2, "PB")])
def test_mix_bn_sum_matches_separate_batchnorms_and_weighted_sum(C, H, N, pattern, dtype, tol):
    """npp_mix_bn_fwd / npp_mix_bn_bwd (the MixedOp edge: sum_k w_k * BN_k(x_k) with affine-free local BatchNorms, 'P' = a plain
    operand) against torch: nn.BatchNorm2d(affine=False) per side + the weighted sum -- output, every dx, dw, running statistics."""
    from npp_amd import _ops as K
    dev = _dev()
    k = len(pattern)
    xs_cpu = [_rand((N, C, H, H), 40 + i) * (1 + 0.3 * i) + 0.2 * i for i in range(k)]
    if dtype == torch.bfloat16:
        xs_cpu = [x.bfloat16().float() for x in xs_cpu]
    w_cpu = torch.softmax(_rand((k,), 60), 0)
    gy = _rand((N, C, H, H), 61)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().float()
    ref_bns = [torch.nn.BatchNorm2d(C, affine=False, momentum=0.1) if ch == "B" else None for ch in pattern]
    xr = [x.clone().requires_grad_(True) for x in xs_cpu]
    wr = w_cpu.clone().requires_grad_(True)
    yr = sum(wr[i] * (ref_bns[i](xr[i]) if ref_bns[i] is not None else xr[i]) for i in range(k))
    yr.backward(gy)
    bns = [torch.nn.BatchNorm2d(C, affine=False, momentum=0.1).to(dev) if ch == "B" else None for ch in pattern]
    xs = [_to_dev(x, dtype).detach().requires_grad_(True) for x in xs_cpu]
    w = w_cpu.to(dev).requires_grad_(True)
    assert K.MIX_FUSE
    y = K.mix_bn_sum(w, [K.BnSide(x, bn, None, private=True) for x, bn in zip(xs, bns)], True)
    y.backward(_to_dev(gy, dtype))
    torch.cuda.synchronize()
    assert rel_err(y.detach().float().cpu().numpy(), yr.detach().numpy()) < tol
    assert rel_err(w.grad.cpu().numpy(), wr.grad.numpy()) < tol * 5
    for i in range(k):
        assert rel_err(xs[i].grad.float().cpu().numpy(), xr[i].grad.numpy()) < tol * 5, i
        if bns[i] is not None:
            assert rel_err(bns[i].running_mean.cpu().numpy(), ref_bns[i].running_mean.numpy()) < tol
            assert rel_err(bns[i].running_var.cpu().numpy(), ref_bns[i].running_var.numpy()) < tol
            assert int(bns[i].num_batches_tracked) == 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,H,pad", [(16, 12, 0), (8, 6, 8), (4, 6, 0), (32, 5, 0)])
def test_interleave2_is_the_channel_shuffle_of_the_concatenation(C, H, pad, dtype):
    """npp_interleave2 (model_search_interact.py:22-36,71-72: channel_shuffle(cat([a, b], 1), groups=2)) and its backward, scalar
    kernel (f32, channel counts not divisible by 8, unaligned rows) and the 16-byte bf16 kernel: pure data movement, bit-exact."""
    from npp_amd import _ops as K
    a_cpu, b_cpu, g_cpu = _rand((2, C, H, H), 70), _rand((2, C, H, H), 71), _rand((2, 2 * C, H, H), 72)
    if dtype == torch.bfloat16:
        a_cpu, b_cpu, g_cpu = a_cpu.bfloat16().float(), b_cpu.bfloat16().float(), g_cpu.bfloat16().float()
    ar, br = a_cpu.clone().requires_grad_(True), b_cpu.clone().requires_grad_(True)
    cat = torch.cat([ar, br], 1)
    n, c2, h, w = cat.shape
    yr = cat.view(n, 2, c2 // 2, h, w).transpose(1, 2).reshape(n, c2, h, w)
    yr.backward(g_cpu)
    a = _to_dev(a_cpu, dtype, slice_pad=pad).detach().requires_grad_(True)
    b = _to_dev(b_cpu, dtype).detach().requires_grad_(True)
    y = K.interleave2(a, b)
    y.backward(_to_dev(g_cpu, dtype))
    torch.cuda.synchronize()
    assert torch.equal(y.detach().float().cpu(), yr.detach())
    assert torch.equal(a.grad.float().cpu(), ar.grad) and torch.equal(b.grad.float().cpu(), br.grad)


def test_add_n_and_fanout_gradient_accumulation():
    """npp_add_n (strided sources, 2..8 terms) and the _FanOut node: a tensor with four consumers gets the same gradient
    as with the autograd engine's own accumulation."""
    from npp_amd import _ops as K
    dev = _dev()
    for dtype, tol in ((torch.float32, 1e-6), (torch.bfloat16, 1e-2)):
        srcs_cpu = [_rand((2, 24, 9, 11), 10 + i) for i in range(5)]
        srcs = [_to_dev(t, dtype, slice_pad=8 if i % 2 else 0) for i, t in enumerate(srcs_cpu)]
        for n in (2, 3, 5):
            out = K.add_n(srcs[:n])
            ref = sum(t.to(torch.bfloat16).float() if dtype == torch.bfloat16 else t for t in srcs_cpu[:n])
            assert rel_err(out.float().cpu().numpy(), ref.numpy()) < tol
    # fan-out: y = pool(x') + pool(x') + bilinear(x') + x'  with x' = 2*x (so that x' has a grad_fn)
    assert K.FANOUT
    x_cpu = _rand((2, 32, 12, 12), 3)
    g_cpu = _rand((2, 32, 12, 12), 4)

    def run(fan):
        K.FANOUT = fan
        K.fan_reset()
        x = _to_dev(x_cpu, torch.float32).detach().requires_grad_(True)
        xp = K.add(x, x)
        a, _ = K.pool3x3(xp, is_avg=True)
        b, _ = K.pool3x3(xp, is_avg=False)
        c = K.bilinear(K.bilinear(xp, 24, 24), 12, 12)
        y = K.add(K.add(a, b), K.add(c, xp))
        y.backward(_to_dev(g_cpu, torch.float32))
        torch.cuda.synchronize()
        return x.grad.float().cpu().numpy()
    try:
        g_fan, g_eng = run(True), run(False)
    finally:
        K.FANOUT = True
        K.fan_reset()
    assert rel_err(g_fan, g_eng) < 1e-5


def test_concat_and_slice_views_roundtrip():
    from npp_amd import _ops as K
    xs = [_rand((2, c, 8, 8), 20 + i) for i, c in enumerate((8, 16, 8, 32))]
    dx = [_to_dev(x, torch.float32).requires_grad_(True) for x in xs]
    y = K.concat(dx)
    ref = torch.cat(xs, dim=1)
    assert rel_err(y.detach().cpu().numpy(), ref.numpy()) == 0.0
    gy = _rand(tuple(ref.shape), 30)
    y.backward(_to_dev(gy, torch.float32))
    off = 0
    for t, x in zip(dx, xs):
        c = x.shape[1]
        assert rel_err(t.grad.cpu().numpy(), gy[:, off:off + c].numpy()) == 0.0
        off += c


def test_kth_smallest_is_exact():
    """OHEM threshold (criterion.py:66-68): radix select must return the identical float, ties included."""
    import ctypes as C
    from npp_amd._lib import lib, check, stream_ptr
    dev = _dev()
    rng = np.random.default_rng(0)
    vals = rng.random(200001).astype(np.float32)
    vals[::7] = -1.0                      # ignored pixels
    vals[5::11] = vals[5]                 # many exact ties
    v = torch.from_numpy(vals).to(dev)
    valid = np.sort(vals[vals >= 0])
    for k in (0, 1, 1000, 131072, len(valid) - 1, 10 ** 7):
        ws = torch.empty(260, dtype=torch.int32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        check(lib().npp_kth_smallest(v.data_ptr(), v.numel(), k, ws.data_ptr(), out.data_ptr(), stream_ptr()))
        torch.cuda.synchronize()
        assert float(out[0]) == float(valid[min(k, len(valid) - 1)]), k
        assert int(out[1]) == len(valid)


def test_fused_adam_matches_torch_adam():
    """npp_amd.optim.FusedAdam (one launch over a device job table) vs torch.optim.Adam on ragged tensor sizes, two
    parameter groups with their own lr / weight decay, several steps."""
    from npp_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    shapes = [(64, 32, 3, 3), (7,), (1,), (4099,), (128, 128), (33, 5)]
    pa = [torch.randn(s, device=dev).requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    groups = lambda ps: [{"params": ps[:3], "lr": 1e-2, "weight_decay": 1e-3}, {"params": ps[3:], "lr": 3e-3}]
    oa = torch.optim.Adam(groups(pa), betas=(0.9, 0.99), eps=1e-8)
    ob = FusedAdam(groups(pb), betas=(0.9, 0.99), eps=1e-8)
    for it in range(4):
        grads = [torch.randn(s, device=dev) for s in shapes]
        for p, q, g in zip(pa, pb, grads):
            p.grad = g.clone()
            q.grad = g.clone()
        oa.step()
        ob.step()
    torch.cuda.synchronize()
    assert ob.device_step_count() == 4
    for p, q in zip(pa, pb):
        assert torch.allclose(p, q, rtol=2e-6, atol=2e-7), float((p - q).abs().max())


def test_fused_adam_checkpoint_round_trip():
    """state_dict() / load_state_dict() carry the step count (augment_lip_sync.py:235,268-278 resume the optimizer): a
    FusedAdam resumed from its own checkpoint, a FusedAdam resumed from torch.optim.Adam's, and torch.optim.Adam resumed from
    FusedAdam's all continue on the trajectory of an uninterrupted torch.optim.Adam."""
    import copy
    from npp_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    shapes = [(32, 16, 3, 3), (5,), (300,)]
    grads = [[torch.randn(s, device=dev) for s in shapes] for _ in range(7)]
    init = [torch.randn(s, device=dev) for s in shapes]

    def fresh(cls):
        ps = [t.clone().requires_grad_(True) for t in init]
        return ps, cls(ps, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-3)

    def run(ps, opt, its):
        for it in its:
            for p, g in zip(ps, grads[it]):
                p.grad = g.clone()
            opt.step()

    p_ref, o_ref = fresh(torch.optim.Adam)
    run(p_ref, o_ref, range(7))
    p_t, o_t = fresh(torch.optim.Adam)
    run(p_t, o_t, range(4))
    p_f, o_f = fresh(FusedAdam)
    run(p_f, o_f, range(4))
    sd_f, sd_t = copy.deepcopy(o_f.state_dict()), copy.deepcopy(o_t.state_dict())
    assert all(float(st["step"]) == 4.0 for st in sd_f["state"].values())       # not the placeholder 0
    cases = []
    for cls, sd, src in ((FusedAdam, sd_f, p_f), (FusedAdam, sd_t, p_t), (torch.optim.Adam, sd_f, p_f)):
        ps = [t.detach().clone().requires_grad_(True) for t in src]
        opt = cls(ps, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-3)
        opt.load_state_dict(copy.deepcopy(sd))
        run(ps, opt, range(4, 7))
        cases.append((cls.__name__, ps, opt))
    torch.cuda.synchronize()
    assert cases[0][2].device_step_count() == 7 and cases[1][2].device_step_count() == 7
    for name, ps, _ in cases:
        for p, q in zip(p_ref, ps):
            assert torch.allclose(p, q, rtol=5e-6, atol=5e-7), (name, float((p - q).abs().max()))
    # load_state_dict AFTER a step: the job table must follow the replaced moment tensors
    o_f.load_state_dict(copy.deepcopy(sd_t))
    run(p_f, o_f, range(4, 7))
    # (p_f had already taken steps 0-3 itself, so it is the same trajectory again)
    for p, q in zip(p_ref, p_f):
        assert torch.allclose(p, q, rtol=5e-6, atol=5e-7), float((p - q).abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_parsing_tta_confusion_matches_reference(dtype):
    """validate_sync's parsing path on the device vs the reference's get_confusion_matrix (tests/golden/eval_parsing.npz):
    integer counts, bit-exact in f32 (a pixel whose top-2 margin is at rounding level may legitimately flip)."""
    from helpers import load_golden
    from npp_amd.evaluate import ParsingConfusion
    from npp_amd import _ops as K
    g = load_golden("eval_parsing.npz")
    dev = torch.device("cuda:0")
    pred = K.cast(torch.from_numpy(g["pred"]).to(dev).contiguous(memory_format=torch.channels_last), dtype)
    flip = K.cast(torch.from_numpy(g["flip"]).to(dev).contiguous(memory_format=torch.channels_last), dtype)
    label = torch.from_numpy(g["label"].astype(np.int64)).to(dev)
    cm = ParsingConfusion(20, 255)
    cm.update(pred[:1], flip[:1], label[:1])        # two batches: counts accumulate
    cm.update(pred[1:], flip[1:], label[1:])
    got = cm.matrix().numpy()
    ref = g["confusion"]
    assert got.sum() == ref.sum()                   # every non-ignored pixel counted once
    if dtype == torch.float32:
        assert np.abs(got - ref).sum() <= 2 * int(g["near_ties"]), np.abs(got - ref).sum()
    else:
        assert np.abs(got - ref).sum() <= 0.02 * ref.sum()     # bf16 logits: ~1 % of arg-maxes move
    cm2 = ParsingConfusion(20, 255)
    cm2.update(pred, None, label)
    if dtype == torch.float32:
        assert np.abs(cm2.matrix().numpy() - g["confusion_noflip"]).sum() <= 2
    true_swap = ParsingConfusion(20, 255, alias_swap=False)
    true_swap.update(pred, flip, label)
    assert not np.array_equal(true_swap.matrix().numpy(), got)


def test_packed_weights_follow_silent_parameter_updates():
    """torch.optim.Adam(fused=True) updates parameters without bumping Tensor._version: the MFMA operand images must
    still be rebuilt for the next training forward, for a free-standing op and for a whole Network."""
    from npp_amd.operations import OPS
    dev = torch.device("cuda:0")
    m = OPS['std_conv_3x3'](32, 1, True).to(dev).train()
    x = torch.randn(2, 32, 12, 12, device=dev).contiguous(memory_format=torch.channels_last)
    y0 = m(x).detach().clone()
    w = m.net[1].weight
    v = w._version
    opt = torch.optim.Adam(m.parameters(), lr=0.5, fused=True)
    m(x).sum().backward()
    opt.step()                       # big step, no version bump
    assert w._version == v
    y1 = m(x).detach()
    assert float((y1 - y0).abs().max()) > 1e-3        # the forward saw the new weights


def test_batched_weight_packing_equals_the_per_weight_kernel():
    """K.WeightPacker (every operand image of a model in ONE launch; tiles of 2048 / 64 x 64 / 256 x taps elements) against
    npp_pack_weight (one thread per element), forward and data-gradient images, bf16 and f32, bit for bit -- shapes cover the
    ragged edges of every tile form: cin not a multiple of 8 (scalar cast), < 256 (several rows per block), > 256, 5x5 / 7x7."""
    import ctypes as C
    from npp_amd import _ops as K
    from npp_amd._lib import lib, check, npp_dtype
    dev = _dev()
    shapes = [(32, 32, 3), (64, 64, 1), (20, 256, 1), (6, 384, 3), (128, 384, 3), (100, 70, 3), (7, 5, 1), (130, 100, 1), (512, 1024, 1),
              (48, 3, 3), (16, 24, 5), (8, 3, 7), (33, 300, 3), (256, 16, 1), (1, 40, 3)]
    g = torch.Generator().manual_seed(3)
    ws = [torch.randn(co, ci, k, k, generator=g).to(dev) for co, ci, k in shapes]
    for dtype in (torch.bfloat16, torch.float32):
        packer = K.WeightPacker(ws)
        packer.pack_if_stale(dtype, dev, force=True)
        torch.cuda.synchronize()
        assert packer.outs is not None and len(packer.outs) == 2 * len(ws)
        for i, w in enumerate(ws):
            co, ci, kh, kw = w.shape
            for dg in (0, 1):
                n = int(lib().npp_packed_weight_elems(co, ci, kh, kw, dg))
                ref = torch.zeros(n, dtype=dtype, device=dev)
                check(lib().npp_pack_weight(w.data_ptr(), co, ci, kh, kw, dg, npp_dtype(dtype), ref.data_ptr(), K.stream_ptr()))
                torch.cuda.synchronize()
                got = packer.outs[2 * i + dg]
                assert torch.equal(got.view(torch.int16 if dtype == torch.bfloat16 else torch.int32),
                                   ref.view(torch.int16 if dtype == torch.bfloat16 else torch.int32)), (tuple(w.shape), dg, dtype)


def test_concat_buffer_parts_written_in_place():
    """K.ConcatBuffer (torch.cat of a cell's node outputs, model_augment.py:62): the producers write their channel slices,
    no copy launch; values and gradients equal torch.cat over separately produced parts, also when a part is consumed again."""
    import torch.nn as nn
    from npp_amd import _ops as K
    dev = _dev()
    torch.manual_seed(0)
    n, c, h, w = 2, 16, 9, 7
    bns = [nn.BatchNorm2d(c).to(dev).train() for _ in range(3)]
    for bn in bns:
        bn.weight.data.uniform_(0.5, 1.5)
        bn.bias.data.normal_(0, 0.1)
    xs = [torch.randn(n, c, h, w, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(4)]

    def run(inplace):
        K.fan_reset()
        ins = [x.clone().requires_grad_(True) for x in xs]
        cb = K.ConcatBuffer(3) if inplace else None
        parts = []
        for k in range(3):
            sa = K.BnSide(ins[k] * 1.0, bns[k], None)             # batch statistics computed on demand
            sb = K.BnSide(ins[3] if k != 1 else parts[0])         # plain side; part 0 is consumed again by part 1
            parts.append(K.bn_add(sa, sb, relu=(k == 2), training=True, out=cb.slot(k) if cb else None))
        y = cb.result(parts) if cb else K.concat(parts)
        for bn in bns:
            bn.zero_grad()
        (y.float() * torch.arange(y.numel(), device=dev).view_as(y).float().cos()).sum().backward()
        return y.detach().clone(), [t.grad.clone() for t in ins], [bn.weight.grad.clone() for bn in bns], parts

    y0, g0, w0, _ = run(False)
    y1, g1, w1, parts = run(True)
    assert parts[1].data_ptr() == parts[0].data_ptr() + c * parts[0].element_size()      # really in place
    assert torch.equal(y0, y1)
    for a, b in zip(g0 + w0, g1 + w1):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5
    ref = torch.cat([p.detach() for p in parts], 1)
    assert torch.equal(ref, y1)


def test_relu_bit_masks_equal_the_tensor_mask():
    """The ReLU bit-mask written by the producer of a tensor (npp_affine_add_m / npp_concat_m, NPP_MASK8) and read by the data
    gradient of its `ReLU -> conv` consumers (conv_g4 / conv_g8) against the bf16 tensor itself as the mask: bit-identical
    gradients, for a standalone tensor, for the parts and the whole of an in-place concatenation and for npp_concat; shapes on
    kernels without bit-mask support fall back silently."""
    from npp_amd import _ops as K
    from npp_amd.model_augment import set_compute_dtype
    from npp_amd.operations import ReLUConvBN
    dev = torch.device("cuda:0")
    set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(3)
        N, C, H = 4, 64, 24
        pre = [ReLUConvBN(C, C, 1, 1, 0).to(dev).train() for _ in range(3)]
        cons_part = ReLUConvBN(C, C, 3, 1, 1).to(dev).train()            # consumes ONE part of the concatenation (conv_g4)
        cons_all = ReLUConvBN(3 * C, 2 * C, 1, 1, 0).to(dev).train()     # consumes the whole buffer
        cons_cat = ReLUConvBN(2 * C, C, 3, 1, 1).to(dev).train()         # consumes an npp_concat result
        cons_s2 = ReLUConvBN(C, C, 3, 2, 1).to(dev).train()              # stride 2: generic kernel, falls back to the tensor mask
        x0 = torch.randn(N, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)

        def run(bits):
            K.RELU_BITS = bits
            K.MASK_STATS[0] = K.MASK_STATS[1] = K.MASK_STATS[2] = 0
            K.fan_reset()
            x = K.cast(x0, torch.bfloat16).detach().requires_grad_(True)
            cb = K.ConcatBuffer(3)
            parts = [pre[i](x, out=cb.slot(i)) for i in range(3)]
            whole = cb.result(parts)
            lone = pre[0](x)                                               # standalone bn_add output
            cat = K.concat([parts[1], lone])
            outs = [cons_part(parts[0]), cons_all(whole), cons_cat(cat), cons_s2(lone), cons_part(lone)]
            loss = sum((o.float() * (i + 1)).sum() for i, o in enumerate(outs))
            for m in pre + [cons_part, cons_all, cons_cat, cons_s2]:
                m.zero_grad()
            loss.backward()
            K.fan_reset()
            torch.cuda.synchronize()
            grads = [x.grad.detach().float().clone()] + [p.grad.detach().clone() for m in pre for p in m.parameters()]
            return grads, tuple(K.MASK_STATS[:2])
        try:
            ref, st0 = run(False)
            got, st1 = run(True)
        finally:
            K.RELU_BITS = True
        assert st0 == (0, 0)
        assert st1[0] >= 4 and st1[1] >= 1, st1          # bit path taken by the stride-1 consumers, refused by the stride-2 one
        for a, b in zip(ref, got):
            # identical masks -> identical data gradients; the weight gradients behind them differ only by atomics order
            assert float((a - b).abs().max()) <= 1e-3 * float(a.abs().max()) + 1e-6
        assert torch.equal(ref[0], got[0]) or float((ref[0] - got[0]).abs().max()) <= 2e-2 * float(ref[0].abs().max())
    finally:
        set_compute_dtype(torch.float32)
