"""Graph-replayed timing of the stencil kernels (max-pool 3x3 + BN statistics, its backward, depthwise dilated 3x3 forward / data
gradient) through the autograd wrappers' C-ABI calls at the network's tensor sizes: us and TB/s of (input + output) bytes."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
from npp_amd._lib import lib, check, geom
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
    x, y, dy, dx = T(c, h), T(c, h), T(c, h), T(c, h)
    mb = x.numel() * 2 / 1e6
    am = torch.empty(N * h * h * c, dtype=torch.uint8, device=dev)
    st = torch.zeros(16 * 2 * c, dtype=torch.float64, device=dev)
    t_pf = timeit(lambda: check(L.npp_pool3x3_fwd(K._byref(x), K._byref(y), am.data_ptr(), 0, 1, st.data_ptr(), s()), "pf"))
    t_pb = timeit(lambda: check(L.npp_pool3x3_bwd(K._byref(dy), am.data_ptr(), K._byref(dx), 0, 1, s()), "pb"))
    out = f"C={c:4d} {h:3d}^2 {mb:6.1f} MB | pool fwd {t_pf:6.1f} us {2*mb/t_pf:5.2f} TB/s | pool bwd {t_pb:6.1f} us {2*mb/t_pb:5.2f}"
    for d in (2, 4):
        w = torch.randn(c, 1, 3, 3, device=dev)
        g = geom(3, 3, 1, 1, d, d, d, d, 1, 1)
        t_df = timeit(lambda: check(L.npp_dwconv_fwd(K._byref(x), w.data_ptr(), K._byref(y), C.byref(g), s()), "df"))
        t_db = timeit(lambda: check(L.npp_dwconv_bwd_data(K._byref(dy), w.data_ptr(), K._byref(x), K._byref(dx), C.byref(g), s()), "db"))
        out += f" | dw d{d} fwd {t_df:6.1f} us {2*mb/t_df:5.2f} bwd {t_db:6.1f} us {3*mb/t_db:5.2f}"
    print(out, flush=True)
