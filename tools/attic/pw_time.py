"""Memory-bound 1x1 convs of the network (fwd shapes and the shapes of their data gradients), graph-replayed over 4 rotating
buffers (151 MB+ working set per shape so that neither L2 nor the Infinity Cache holds the operands): us and HBM TB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
dev = torch.device("cuda:0")
N, H, iters, NB = 16, 96, 16, 4
for cin, cout, relu, stats in [(128, 128, 0, 0), (128, 128, 1, 0), (128, 128, 0, 1), (512, 128, 0, 0), (128, 128, 1, 1), (512, 128, 1, 1), (128, 512, 0, 0), (384, 128, 1, 1), (128, 384, 0, 0), (256, 128, 1, 1)]:
    xs = [K.cast(torch.randn(N, cin, H, H, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16) for _ in range(NB)]
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
    for x in xs[:2]:
        K.conv2d(x, w, None, 1, 0, 1, relu_in=bool(relu), want_stats=bool(stats))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    keep = []
    with torch.cuda.graph(g):
        for i in range(iters):
            keep.append(K.conv2d(xs[i % NB], w, None, 1, 0, 1, relu_in=bool(relu), want_stats=bool(stats)))
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / iters)
    mb = N * H * H * (cin + cout) * 2 / 1e6
    print(f"{cin:4d}->{cout:4d} relu={relu} stats={stats}: {best:7.1f} us  {mb:6.1f} MB  {mb / best:5.2f} TB/s", flush=True)
    del g, keep, xs
