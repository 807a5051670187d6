"""max/avg pool 3x3 fwd + bwd + fused statistics vs torch on odd shapes (edge maps of the tiny networks)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from npp_amd import _ops as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
bad = 0
for dtype in (torch.float32, torch.bfloat16):
    for (C, H, W, s, avg) in itertools.product((8, 16, 40), (2, 3, 5, 16), (2, 3, 7, 16), (1, 2), (False, True)):
        N = 2
        x_cpu = torch.randn(N, C, H, W)
        if dtype == torch.bfloat16:
            x_cpu = x_cpu.bfloat16().float()
        xr = x_cpu.clone().requires_grad_(True)
        yr = F.avg_pool2d(xr, 3, s, 1, count_include_pad=False) if avg else F.max_pool2d(xr, 3, s, 1)
        gy = torch.randn_like(yr)
        if dtype == torch.bfloat16:
            gy = gy.bfloat16().float()
        yr.backward(gy)
        x = x_cpu.to(dev).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y, st = K.pool3x3(x, avg, s, want_stats=True)
        y.backward(gy.to(dev).to(dtype).contiguous(memory_format=torch.channels_last))
        torch.cuda.synchronize()
        tol = 1e-5 if dtype == torch.float32 else 1e-2
        e1 = (y.detach().float().cpu() - yr.detach()).abs().max().item()
        e2 = (x.grad.float().cpu() - xr.grad).abs().max().item()
        yf = y.detach().double()
        ref = torch.cat([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))]).cpu()
        got = st.view(-1, 2 * C).sum(0).cpu()
        e3 = ((got - ref).abs() / (ref.abs() + 1e-6)).max().item()
        if e1 > tol or e2 > tol * 4 or e3 > 1e-5:
            bad += 1
            print("MISMATCH", dtype, C, H, W, s, "avg" if avg else "max", e1, e2, e3)
print("checked, mismatches:", bad)
