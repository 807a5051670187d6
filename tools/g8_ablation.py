"""conv_g8 ablation timings: one process per NPP_G8_DBG value (the switch is read once per process).
Prints one line per (shape, direction): graph-replayed us per launch of the bare 1x1 forward (no input ReLU, no statistics) and of the
data gradient (bf16 ReLU mask), N = 16, 96 x 96, bf16.  DBG bits: 1 no epilogue, 2 no MFMA, 4 no DMA, 8 no fragment reads."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K

dev = torch.device("cuda:0")
N, H, iters = 16, 96, 10
dbg = os.environ.get("NPP_G8_DBG", "0")


def timed(launch):
    for _ in range(2):
        launch()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            launch()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / iters)
    return best


# clocks up before the first measurement (the first shape of a cold process read 15 % slow)
_xw = K.cast(torch.randn(N, 1024, H, H, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
_ww = torch.randn(512, 1024, 1, 1, device=dev) * 0.05
for _ in range(300):
    K.conv2d(_xw, _ww, None, 1, 0, 1, relu_in=False, want_stats=False)
torch.cuda.synchronize()
del _xw, _ww

for cin, cout in ((1024, 512), (1024, 384), (512, 256)):
    x = K.cast(torch.randn(N, cin, H, H, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
    us = timed(lambda: K.conv2d(x, w, None, 1, 0, 1, relu_in=False, want_stats=False))
    gf = 2.0 * N * H * H * cout * cin / 1e9
    print(f"dbg={dbg} fwd   {cin}->{cout} {us:8.1f} us {gf / us * 1e3:7.0f} TF/s", flush=True)
    gy = K.cast(torch.randn(N, cout, H, H, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
    dx = K.new_nhwc(N, cin, H, H, x.dtype, x.device)
    geo = K.geom(1, 1, 1, 1, 0, 0, 1, 1, (1, 1), 0)
    wp = K.packed_weight(w, True, x.dtype)
    us = timed(lambda: K._conv_launch(gy, wp.data_ptr(), None, K._byref(x), dx, None, geo, K.stream_ptr(), "dgrad"))
    print(f"dbg={dbg} dgrad {cin}->{cout} {us:8.1f} us {gf / us * 1e3:7.0f} TF/s", flush=True)
    if dbg == "0":      # the in-model form of the forward: input ReLU + BatchNorm statistics epilogue
        us = timed(lambda: K.conv2d(x, w, None, 1, 0, 1, relu_in=True, want_stats=True))
        print(f"dbg=full fwd   {cin}->{cout} {us:8.1f} us {gf / us * 1e3:7.0f} TF/s  (input ReLU + statistics)", flush=True)
