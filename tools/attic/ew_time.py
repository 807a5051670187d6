"""Graph-replayed timing of the memory-bound kernels through the C ABI at the network's tensor sizes: achieved TB/s."""
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
for c, h in [(32, 96), (128, 96), (64, 48), (256, 48), (128, 24), (256, 12), (512, 96)]:
    a, b, o = T(c, h), T(c, h), T(c, h)
    mb = a.numel() * 2 / 1e6
    ss = torch.randn(2 * c, device=dev)
    mi = torch.cat([torch.zeros(c, device=dev), torch.ones(c, device=dev)])
    nb = L.npp_reduce_blocks(N * h * h, c, 1 if True else 0) if False else L.npp_reduce_blocks(N * h * h, c, K.L.npp_dtype(a.dtype))
    sums = torch.empty(nb * 2 * c, dtype=torch.float64, device=dev)
    co = torch.randn(3 * c, device=dev)
    s = K.stream_ptr
    t_aff = timeit(lambda: check(L.npp_affine_add(K._byref(o), K._byref(a), ss.data_ptr(), K._byref(b), ss.data_ptr(), 0, s()), "aff"))
    t_red = timeit(lambda: check(L.npp_bn_bwd_reduce(K._byref(a), K._byref(b), None, mi.data_ptr(), sums.data_ptr(), nb, s()), "red"))
    t_app = timeit(lambda: check(L.npp_bn_bwd_apply(K._byref(a), K._byref(b), None, co.data_ptr(), K._byref(o), s()), "app"))
    dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
    gam = torch.ones(c, device=dev)
    t_co = timeit(lambda: check(L.npp_bn_bwd_coeffs(sums.data_ptr(), nb, float(N * h * h), mi.data_ptr(), gam.data_ptr(), co.data_ptr(), dg.data_ptr(), db.data_ptr(), c, s()), "co"))
    t_add = timeit(lambda: K.add_n([a, b]))
    t_cpy = timeit(lambda: check(L.npp_copy(K._byref(a), K._byref(o), s()), "cpy"))
    print(f"C={c:4d} {h:3d}^2 {mb:6.1f} MB | affine_add {t_aff:6.1f} us {3*mb/t_aff:5.2f} TB/s | bwd_reduce {t_red:6.1f} us {2*mb/t_red:5.2f} | "
          f"coeffs(nb={nb}) {t_co:5.1f} us | bwd_apply {t_app:6.1f} us {3*mb/t_app:5.2f} | add_n(2) {t_add:6.1f} us {3*mb/t_add:5.2f} | copy {t_cpy:6.1f} us {2*mb/t_cpy:5.2f}", flush=True)
