"""Graph-replayed timing of npp_bilinear_bwd / fwd at the network's shapes (N = 16, bf16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
from npp_amd._lib import lib, check
dev = torch.device("cuda:0")
N, iters = 16, 20
def T(c, h):
    return K.cast(torch.randn(N, c, h, h, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / iters)
    return best
L = lib()
for c, lo, hi in [(128, 48, 96), (128, 24, 96), (128, 12, 96), (256, 24, 48), (256, 12, 24), (512, 12, 24), (64, 48, 96), (24, 96, 384)]:
    try:
        x, y = T(c, lo), T(c, hi)
        s = K.stream_ptr
        tb = timeit(lambda: check(L.npp_bilinear_bwd(K._byref(y), K._byref(x), s()), "bwd"))
        nb = int(L.npp_bilinear_bwd_ws_bytes(K._byref(y), K._byref(x)))
        ts = 0.0
        if nb:
            ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            ts = timeit(lambda: check(L.npp_bilinear_bwd_ws(K._byref(y), K._byref(x), 1, ws.data_ptr(), nb, s()), "bwds"))
        tf = timeit(lambda: check(L.npp_bilinear_fwd(K._byref(x), K._byref(y), s()), "fwd"))
        mb = (x.numel() + y.numel()) * 2 / 1e6
        print(f"C={c:4d} {lo:3d}->{hi:3d}  {mb:7.1f} MB | bwd {tb:7.1f} us {mb / tb:5.2f} TB/s | separable {ts:7.1f} us | fwd {tf:7.1f} us {mb / tf:5.2f} TB/s", flush=True)
    except Exception as e:
        print(c, lo, hi, "failed", e)
