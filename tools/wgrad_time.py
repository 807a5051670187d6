"""Weight-gradient kernel timing per shape (bf16, direct npp_conv_wgrad launches, HIP events over 20 launches after 3 warm-ups).
    python3 tools/wgrad_time.py            (env: NPP_DISABLE_WG3=1, NPP_WG3_BLOCKS=n, NPP_WG4_BLOCKS=n)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
from npp_amd import _lib as L
from npp_amd._lib import lib, check, desc, geom

dev = torch.device("cuda:0")
SHAPES = [(128, 128, 3, 96, 16), (384, 128, 3, 96, 16), (256, 256, 3, 48, 16), (512, 512, 3, 24, 16), (256, 256, 3, 12, 16),
          (128, 128, 3, 24, 16), (1024, 512, 1, 96, 16), (512, 128, 1, 96, 16), (384, 6, 3, 96, 16), (32, 32, 3, 96, 16), (64, 64, 3, 48, 16)]
for cin, cout, k, hw, n in SHAPES:
    x = K.cast(torch.randn(n, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
    dy = K.cast(torch.randn(n, cout, hw, hw, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
    nel = int(lib().npp_packed_weight_elems(cout, cin, k, k, 0))
    dwp = torch.zeros(nel, dtype=torch.float32, device=dev)
    g = geom(k, k, 1, 1, k // 2, k // 2, 1, 1, 1, 1)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        check(lib().npp_conv_wgrad(C.byref(desc(x)), C.byref(desc(dy)), dwp.data_ptr(), C.byref(g), s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        check(lib().npp_conv_wgrad(C.byref(desc(x)), C.byref(desc(dy)), dwp.data_ptr(), C.byref(g), s))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    fl = 2.0 * n * hw * hw * cin * cout * k * k
    print(f"{cin:5d}->{cout:4d} k{k} {hw:3d}^2 N={n}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
