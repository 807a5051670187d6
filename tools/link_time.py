"""What does one link of a dependent kernel chain cost inside a replayed hipGraph?  Chains of 200 dependent launches on one stream:
npp_stamp (a 1-thread kernel: the launch floor), and the fused add on NHWC bf16 tensors of the encoder's map sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
from npp_amd._lib import lib, check
dev = torch.device("cuda:0")
n = 200


def timed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / n)
    return best


buf = torch.zeros(512, dtype=torch.int64, device=dev)
def stamps():
    for i in range(n):
        check(lib().npp_stamp(buf.data_ptr(), i % 512, K.stream_ptr()), "stamp")
print(f"npp_stamp chain: {timed(stamps):6.2f} us per link", flush=True)
for c, h in [(256, 12), (128, 24), (64, 48), (32, 96), (128, 96)]:
    a = K.cast(torch.randn(16, c, h, h, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
    b = K.cast(torch.randn(16, c, h, h, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
    def chain():
        x = a
        with torch.no_grad():
            for i in range(n):
                x = K.add(x, b)
        return x
    us = timed(chain)
    mb = a.numel() * 2 * 3 / 1e6
    print(f"add chain C={c:4d} {h:3d}^2 ({mb:6.1f} MB per link): {us:6.2f} us per link = {mb / us:5.2f} TB/s", flush=True)
