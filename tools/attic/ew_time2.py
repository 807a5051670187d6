"""More memory-bound kernels at network sizes (graph-replayed): two-sided BN backward, SE scale / reduce, depthwise, pool, bilinear, stats."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
from npp_amd._lib import lib, check
dev = torch.device("cuda:0")
N, iters = 16, 20
def T(c, h):
    return K.cast(torch.randn(N, c, h, h, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / iters)
    return best
L = lib()
s = K.stream_ptr
for c, h in [(32, 96), (128, 96), (64, 48), (128, 24), (256, 12)]:
    a, b, d, o1, o2 = T(c, h), T(c, h), T(c, h), T(c, h), T(c, h)
    mb = a.numel() * 2 / 1e6
    mi = torch.cat([torch.zeros(c, device=dev), torch.ones(c, device=dev)])
    co = torch.randn(3 * c, device=dev)
    nb = L.npp_reduce_blocks(N * h * h, c, K.L.npp_dtype(a.dtype))
    sums = torch.empty(nb * 3 * c, dtype=torch.float64, device=dev)
    gate = torch.rand(N * c, device=dev)
    pooled = torch.zeros(N * c, device=dev)
    st = torch.zeros(K.R * 2 * c, dtype=torch.float64, device=dev)
    wdw = torch.randn(c, 1, 3, 3, device=dev)
    gdw = K.geom(3, 3, 1, 1, 2, 2, 2, 2, 1, 1)
    amax = torch.empty(N * h * h * c, dtype=torch.uint8, device=dev)
    t_r2 = timeit(lambda: check(L.npp_bn_bwd_reduce2(K._byref(d), K._byref(a), K._byref(b), None, mi.data_ptr(), mi.data_ptr(), sums.data_ptr(), nb, s()), "r2"))
    t_a2 = timeit(lambda: check(L.npp_bn_bwd_apply2(K._byref(d), K._byref(a), K._byref(b), None, co.data_ptr(), co.data_ptr(), K._byref(o1), K._byref(o2), s()), "a2"))
    t_sc = timeit(lambda: check(L.npp_scale_channels(K._byref(a), gate.data_ptr(), K._byref(o1), s()), "sc"))
    t_sr = timeit(lambda: check(L.npp_se_bwd_reduce(K._byref(d), K._byref(a), pooled.data_ptr(), s()), "sr"))
    t_sa = timeit(lambda: check(L.npp_se_bwd_apply(K._byref(d), gate.data_ptr(), pooled.data_ptr(), K._byref(o1), s()), "sa"))
    t_cs = timeit(lambda: check(L.npp_channel_stats(K._byref(a), st.data_ptr(), s()), "cs"))
    t_dw = timeit(lambda: check(L.npp_dwconv_fwd(K._byref(a), wdw.data_ptr(), K._byref(o1), C.byref(gdw), s()), "dw"))
    t_bl = timeit(lambda: K.bilinear(a, 2 * h, 2 * h)) if h <= 48 else 0.0
    print(f"C={c:4d} {h:3d}^2 {mb:6.1f} MB | reduce2 {t_r2:6.1f} us {3*mb/t_r2:5.2f} TB/s | apply2 {t_a2:6.1f} {5*mb/t_a2:5.2f} | scale_ch {t_sc:6.1f} {2*mb/t_sc:5.2f} | "
          f"se_bwd_red {t_sr:6.1f} {2*mb/t_sr:5.2f} | se_bwd_app {t_sa:6.1f} {2*mb/t_sa:5.2f} | stats {t_cs:6.1f} {mb/t_cs:5.2f} | dw3x3d2 {t_dw:6.1f} {2*mb/t_dw:5.2f} | bilin x2 {t_bl:6.1f} {5*mb/max(t_bl,1e-9):5.2f}", flush=True)
# bilinear backward (x2 and x4 upsampling gradients)
for c, h, f in [(64, 48, 2), (128, 24, 4), (256, 12, 8), (128, 48, 2)]:
    x = T(c, h).requires_grad_(True)
    y = K.bilinear(x, f * h, f * h)
    gy = T(c, f * h)
    dx = K.new_nhwc(N, c, h, h, x.dtype, dev)
    t_b = timeit(lambda: check(L.npp_bilinear_bwd(K._byref(gy), K._byref(dx), s()), "bb"))
    print(f"bilinear_bwd C={c} {h}->{f*h}: {t_b:6.1f} us  {(gy.numel() + dx.numel()) * 2 / 1e6 / t_b:5.2f} TB/s", flush=True)
