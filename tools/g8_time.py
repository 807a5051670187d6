"""Graph-replayed timing of conv forward launches through the C ABI: python tools/g8_time.py [N] -- one line per shape.
Run with NPP_DISABLE_G8=1 for the conv_s1 numbers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SHAPES = [(1024, 512, 1, 96), (1024, 384, 1, 96), (512, 256, 1, 96), (512, 128, 1, 96), (384, 128, 1, 96), (128, 128, 1, 96),
          (128, 128, 3, 96), (384, 128, 3, 96), (256, 256, 3, 48), (256, 256, 1, 48)]
iters = 20
if os.environ.get('NPP_TIME_SET') == 'ab':
    SHAPES = [(1024, 512, 1, 96), (512, 128, 1, 96), (512, 512, 1, 24)]
if os.environ.get('NPP_TIME_SET') == 'g4c':
    SHAPES = [(1024, 1024, 3, 12), (128, 128, 3, 96), (384, 128, 3, 96), (512, 128, 1, 96), (128, 128, 1, 96), (1024, 512, 1, 96)]
if os.environ.get('NPP_TIME_SET') == 'g4b':
    SHAPES = [(256, 256, 3, 48), (128, 128, 3, 48), (128, 128, 3, 24), (256, 256, 3, 12), (64, 64, 3, 48), (512, 512, 3, 24), (256, 256, 3, 24)]
if os.environ.get('NPP_TIME_SET') == 'c32':
    SHAPES = [(32, 32, 3, 96), (128, 32, 1, 96), (32, 32, 1, 96), (64, 64, 3, 48)]
if os.environ.get('NPP_TIME_SET') == 'g4':
    SHAPES = [(128, 128, 3, 24), (256, 256, 3, 12), (64, 64, 3, 48), (256, 64, 1, 48), (64, 64, 1, 48), (128, 128, 1, 24), (512, 512, 1, 24), (1024, 256, 1, 12), (256, 256, 1, 48)]
if os.environ.get('NPP_TIME_SET') == 'h3':
    SHAPES = [(128, 128, 3, 96), (384, 128, 3, 96), (256, 256, 3, 48), (64, 64, 3, 48), (128, 128, 3, 48), (512, 512, 3, 48)]
if os.environ.get('NPP_TIME_SET') == 'small':
    SHAPES = [(256, 256, 1, 48), (256, 128, 1, 48), (256, 64, 1, 48), (512, 512, 1, 24), (512, 256, 1, 24), (512, 128, 1, 24), (128, 128, 1, 24), (1024, 256, 1, 12)]
if len(sys.argv) > 2 and sys.argv[2] == "one":
    SHAPES = [(1024, 512, 1, 96)]
for cin, cout, k, H in SHAPES:
    for relu in ((False,) if len(sys.argv) > 2 else (True, False)):
        x = K.cast(torch.randn(N, cin, H, H, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        for _ in range(2):
            K.conv2d(x, w, None, 1, k // 2, 1, relu_in=relu, want_stats=relu)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                y, st = K.conv2d(x, w, None, 1, k // 2, 1, relu_in=relu, want_stats=relu)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); g.replay(); e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) * 1e3 / iters)
        gf = 2.0 * N * H * H * cout * cin * k * k / 1e9
        print(f"{cin:5d}->{cout:4d} k{k} {H}^2 relu+stats={int(relu)}: {best:7.1f} us  {gf / best * 1e3:6.0f} TF/s", flush=True)
        del g
