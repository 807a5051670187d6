"""Time one conv shape (fwd) through the C ABI with HIP events: python tools/conv_time.py cin cout k H [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
dev = torch.device("cuda:0")
cin, cout, k, H = [int(a) for a in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
x = K.cast(torch.randn(16, cin, H, H, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
w = torch.randn(cout, cin, k, k, device=dev) * 0.05
for _ in range(3):
    K.conv2d(x, w, None, 1, k // 2, 1, relu_in=True, want_stats=True)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(iters):
        y, st = K.conv2d(x, w, None, 1, k // 2, 1, relu_in=True, want_stats=True)
g.replay(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); g.replay(); e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / iters
gf = 2.0 * 16 * H * H * cout * cin * k * k / 1e9
print(f"{cin}->{cout} k{k} {H}^2: {us:.1f} us  {gf / us * 1e3:.0f} TF/s  env={ {k_: v for k_, v in os.environ.items() if k_.startswith('NPP_')} }")
