"""Graph-replayed timing of conv data-gradient launches (1x1 / 3x3, with the ReLU-backward mask): python tools/g8_time_dgrad.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
dev = torch.device("cuda:0")
N, iters = 16, 10
SHAPES = [(1024, 512, 1, 96), (1024, 384, 1, 96), (512, 256, 1, 96), (512, 128, 1, 96), (384, 128, 1, 96), (128, 128, 1, 96),
          (128, 128, 3, 96), (384, 128, 3, 96)]
if os.environ.get('NPP_TIME_SET') == 'g4b':
    SHAPES = [(256, 256, 3, 48), (128, 128, 3, 48), (128, 128, 3, 24), (256, 256, 3, 12), (64, 64, 3, 48), (512, 512, 3, 24), (256, 256, 3, 24)]
if os.environ.get('NPP_TIME_SET') == 'c32':
    SHAPES = [(32, 32, 3, 96), (128, 32, 1, 96), (32, 32, 1, 96), (64, 64, 3, 48)]
if os.environ.get('NPP_TIME_SET') == 'g4':
    SHAPES = [(128, 128, 3, 24), (256, 256, 3, 12), (64, 64, 3, 48), (256, 64, 1, 48), (64, 64, 1, 48), (128, 128, 1, 24), (512, 512, 1, 24), (1024, 256, 1, 12), (256, 256, 1, 48)]
if os.environ.get('NPP_TIME_SET') == 'small':
    SHAPES = [(256, 256, 1, 48), (256, 128, 1, 48), (256, 64, 1, 48), (512, 512, 1, 24), (512, 256, 1, 24), (512, 128, 1, 24), (128, 128, 1, 24), (1024, 256, 1, 12)]
for cin, cout, k, H in SHAPES:
    x = K.cast(torch.randn(N, cin, H, H, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16).requires_grad_(True)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    x = x.detach()
    gy = K.cast(torch.randn(N, cout, H, H, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
    dx = K.new_nhwc(N, cin, H, H, x.dtype, x.device)
    geo = K.geom(k, k, 1, 1, k - 1 - k // 2, k - 1 - k // 2, 1, 1, (1, 1), 0)
    wp = K.packed_weight(w, True, x.dtype)
    def launch():    # exactly _Conv2d.backward's data-gradient launch
        K._conv_launch(gy, wp.data_ptr(), None, K._byref(x), dx, None, geo, K.stream_ptr(), "npp_conv_fwd(dgrad)")
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
    mb = N * H * H * (cout + 2 * cin) * 2 / 1e6
    print(f"dgrad {cin:5d}->{cout:4d} k{k} {H}^2: {best:7.1f} us  {gf / best * 1e3:6.0f} TF/s  {mb / best:6.2f} TB/s (dy + mask + dx)", flush=True)
    del g
