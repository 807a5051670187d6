"""conv_g8 forward, 1024 -> 512 / 1024 -> 384 / 512 -> 256 1x1 @96^2, N = 16: ReLU on the input and BatchNorm statistics in the epilogue, separately (graph-replayed us)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
dev = torch.device("cuda:0")
N, iters = 16, 20
for cin, cout in [(1024, 512), (1024, 384), (512, 256)]:
    x = K.cast(torch.randn(N, cin, 96, 96, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
    for relu in (False, True):
        for stats in (False, True):
            for _ in range(2):
                K.conv2d(x, w, None, 1, 0, 1, relu_in=relu, want_stats=stats)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(iters):
                    y, st = K.conv2d(x, w, None, 1, 0, 1, relu_in=relu, want_stats=stats)
            g.replay(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); g.replay(); e.record(); torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e) * 1e3 / iters)
            gf = 2.0 * N * 96 * 96 * cout * cin / 1e9
            print(f"{cin:5d}->{cout:4d} relu={int(relu)} stats={int(stats)}: {best:7.1f} us  {gf / best * 1e3:6.0f} TF/s", flush=True)
            del g
