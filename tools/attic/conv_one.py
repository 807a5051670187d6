import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
dev = torch.device("cuda:0")
N = 16
cin, cout, k, H = [int(a) for a in sys.argv[1:5]]
x = K.cast(torch.randn(N, cin, H, H, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
w = torch.randn(cout, cin, k, k, device=dev) * 0.05
for _ in range(3):
    K.conv2d(x, w, None, 1, k // 2, 1, relu_in=True, want_stats=True)
torch.cuda.synchronize()
