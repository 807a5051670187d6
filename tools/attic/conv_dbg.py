import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
from tools.conv_bench import bench
dev = torch.device("cuda:0")
N = 16
for cin, cout, k, H in [(128, 128, 3, 96), (1024, 512, 1, 96), (384, 128, 3, 96), (512, 128, 1, 96)]:
    x = K.cast(torch.randn(N, cin, H, H, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    t = bench(lambda: K.conv2d(x, w, None, 1, k // 2, 1, relu_in=True, want_stats=True), iters=20)
    gf = 2.0 * N * H * H * cout * cin * k * k / 1e9
    print(f"dbg={os.environ.get('NPP_S1_DBG','0')} {cin}->{cout} k{k}: {t:8.1f} us  {gf/t*1e3:7.1f} TF/s")
