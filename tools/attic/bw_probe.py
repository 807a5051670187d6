"""HBM rates of plain streaming kernels on this device (graph-replayed over rotating buffers): fill, copy, read-only sum."""
import torch
dev = torch.device("cuda:0")
NB, iters = 6, 12
n = 16 * 96 * 96 * 512          # 151 MB of bf16
bufs = [torch.empty(n, dtype=torch.bfloat16, device=dev) for _ in range(NB)]
srcs = [torch.randn(n, dtype=torch.float32, device=dev).bfloat16() for _ in range(NB)]
def timed(fn, mb, name):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            fn(i % NB)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / iters)
    print(f"{name:28s} {best:7.1f} us  {mb / best:5.2f} TB/s", flush=True)
mb = n * 2 / 1e6
timed(lambda i: bufs[i].zero_(), mb, "fill 151 MB")
timed(lambda i: bufs[i].copy_(srcs[i]), 2 * mb, "copy 151 MB -> 151 MB")
timed(lambda i: bufs[i].add_(1.0), 2 * mb, "in-place add (r+w same)")
q = n // 4
timed(lambda i: bufs[i][:q].copy_(srcs[i][:q]), 2 * mb / 4, "copy 37.7 MB")
outs = [torch.empty(1, dtype=torch.float32, device=dev) for _ in range(NB)]
timed(lambda i: torch.sum(srcs[i].view(torch.int16), dtype=torch.int64, out=None), mb, "read-only sum 151 MB")
