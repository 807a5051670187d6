"""Graph-replayed timing of conv weight-gradient launches through the C ABI: python tools/g8_time_wgrad.py (NPP_DISABLE_WG4=1 for the
previous kernels)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
from npp_amd._lib import lib, check
dev = torch.device("cuda:0")
N, iters = 16, 10
SHAPES = [(128, 128, 3, 96), (384, 128, 3, 96), (1024, 512, 1, 96), (1024, 384, 1, 96), (512, 256, 1, 96), (512, 128, 1, 96),
          (384, 128, 1, 96), (128, 128, 1, 96), (256, 256, 3, 48), (128, 128, 3, 24), (256, 256, 3, 12), (512, 512, 3, 24),
          (512, 512, 1, 24), (256, 256, 1, 48), (1024, 256, 1, 12)]
if os.environ.get('NPP_TIME_SET') == 'h3':
    SHAPES = [(128, 128, 3, 96), (384, 128, 3, 96), (256, 256, 3, 96), (128, 128, 3, 64)]
if os.environ.get('NPP_TIME_SET') == 'narrow':
    SHAPES = [(64, 64, 3, 48), (32, 32, 3, 96), (128, 32, 1, 96), (256, 64, 1, 48), (64, 64, 1, 48), (32, 32, 1, 96), (64, 128, 3, 96)]
for cin, cout, k, H in SHAPES:
    x = K.cast(torch.randn(N, cin, H, H, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
    gy = K.cast(torch.randn(N, cout, H, H, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
    nel = lib().npp_packed_weight_elems(cout, cin, k, k, 0)
    dwp = torch.zeros(nel, dtype=torch.float32, device=dev)
    geo = K.geom(k, k, 1, 1, k // 2, k // 2, 1, 1, 1, 1)
    nsl = int(lib().npp_conv_wgrad_splits(K._byref(x), K._byref(gy), C.byref(geo))) if k > 1 else 0
    slabs = torch.empty(max(nsl, 1) * nel, dtype=torch.float32, device=dev)
    dw = torch.empty(cout, cin, k, k, dtype=torch.float32, device=dev)
    def launch():      # the whole weight-gradient of the layer: kernel + (for KxK) the unpack into OIHW
        if nsl > 0:
            check(lib().npp_conv_wgrad_slabs(K._byref(x), K._byref(gy), slabs.data_ptr(), nsl, C.byref(geo), K.stream_ptr()), "slabs")
            check(lib().npp_unpack_wgrad_sum(slabs.data_ptr(), nsl, cout, cin, k, k, dw.data_ptr(), K.stream_ptr()), "unpack_sum")
        else:
            check(lib().npp_conv_wgrad(K._byref(x), K._byref(gy), dwp.data_ptr(), C.byref(geo), K.stream_ptr()), "npp_conv_wgrad")
            if k > 1:
                check(lib().npp_unpack_wgrad(dwp.data_ptr(), cout, cin, k, k, dw.data_ptr(), K.stream_ptr()), "unpack")
    for _ in range(2):
        launch()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            launch()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / iters)
    gf = 2.0 * N * H * H * cout * cin * k * k / 1e9
    print(f"wgrad {cin:5d}->{cout:4d} k{k} {H}^2 (slabs {nsl}): {best:7.1f} us  {gf / best * 1e3:6.0f} TF/s", flush=True)
    del g
