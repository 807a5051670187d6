"""relu / stats cost split on two shapes (graph-replayed)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
dev = torch.device("cuda:0")
N, iters = 16, 20
for cin, cout, k, H in [(1024, 512, 1, 96), (128, 128, 3, 96), (512, 128, 1, 96)]:
    for relu, stats in [(0, 0), (1, 0), (0, 1), (1, 1)]:
        x = K.cast(torch.randn(N, cin, H, H, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        for _ in range(2):
            K.conv2d(x, w, None, 1, k // 2, 1, relu_in=bool(relu), want_stats=bool(stats))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                K.conv2d(x, w, None, 1, k // 2, 1, relu_in=bool(relu), want_stats=bool(stats))
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); g.replay(); e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) * 1e3 / iters)
        print(f"{cin}->{cout} k{k} relu={relu} stats={stats}: {best:7.1f} us", flush=True)
