"""Strided / stem conv shapes of the bench model: forward and forward + backward (data + weight gradient) per call, graph-replayed
(20 calls per replay), N = 16 bf16.      python3 tools/strided_time.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K

dev = torch.device("cuda:0")
SHAPES = [(3, 64, 3, 2, 384, False, False), (64, 128, 3, 2, 192, False, True), (64, 64, 3, 2, 96, True, True), (128, 128, 3, 2, 48, True, True),
          (256, 256, 3, 2, 24, True, True), (128, 32, 1, 2, 96, True, True), (256, 64, 1, 2, 48, True, True), (512, 128, 1, 2, 24, True, True),
          (128, 128, 3, 1, 96, True, True)]
import ctypes as C
from npp_amd import _lib as L


def fam_time(fam, fn, n=10):
    """mean GPU microseconds per launch of kernel family `fam` while fn() runs n times (HIP events around every launch)"""
    lib = L.lib()
    lib.npp_prof_begin(L.FAM[fam], L.NPP_BF16)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    ms, fl, by, nl = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
    lib.npp_prof_end(C.byref(ms), C.byref(fl), C.byref(by), C.byref(nl))
    return (ms.value * 1e3 / nl.value if nl.value else 0.0), nl.value // n


for cin, cout, k, s, H, relu, need_dx in SHAPES:
    x = K.cast(torch.randn(16, cin, H, H, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16).requires_grad_(need_dx)
    w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).requires_grad_(True)
    y, _ = K.conv2d(x, w, None, s, k // 2, 1, relu_in=relu, want_stats=True)
    gy = torch.randn_like(y)

    def fwd():
        K.conv2d(x, w, None, s, k // 2, 1, relu_in=relu, want_stats=True)

    def both():
        y, st = K.conv2d(x, w, None, s, k // 2, 1, relu_in=relu, want_stats=True)
        torch.autograd.grad(y, [x, w] if need_dx else [w], gy)

    for _ in range(2):
        both()
    out = []
    for fam in ("conv_igemm", "conv_s1", "conv_g4", "conv_g8"):
        us, nl = fam_time(fam, fwd)
        if nl:
            out.append(f"fwd {fam} {us:7.1f} us")
    for fam in ("conv_igemm", "conv_s1", "conv_g4", "conv_g8", "conv_wgrad"):
        us, nl = fam_time(fam, both)
        if nl:
            out.append(f"fwd+bwd {fam} x{nl} {us:7.1f} us/launch")
    gf = 2.0 * 16 * (H // s) ** 2 * cout * cin * k * k / 1e9
    print(f"{cin:4d}->{cout:4d} k{k} s{s} {H:3d}^2 ({gf:5.1f} GF): " + "   ".join(out), flush=True)
