"""GPU box: the ordered list of device launches of ONE MixedOp (config 5's shape: C = 32, batch 8, 96 x 96) forward + backward."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
from npp_amd.model_augment import set_compute_dtype
from npp_amd.model_search_interact import MixedOp

dev = torch.device("cuda:0")
set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
C, up = int(os.environ.get("C", 32)), (int(os.environ["UP"]) if os.environ.get("UP") else None)
op = MixedOp(C, 1, up).to(dev).train()
hw = int(os.environ.get("HW", 96))
x = torch.randn(8, hw, hw, C, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2).requires_grad_(True)
w = torch.softmax(torch.randn(7, device=dev), 0).requires_grad_(True)


def run():
    K.fan_reset()
    y = op(K.relu(x) if hasattr(K, "relu") else x, w)
    y.float().sum().backward()


for _ in range(3):
    run()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run()
    torch.cuda.synchronize()
evs = sorted((e for e in prof.events() if str(e.device_type).endswith("CUDA")), key=lambda e: e.time_range.start)
for i, e in enumerate(evs):
    print(f"{i:3d} {e.device_time:7.1f} us  {e.name.replace('void (anonymous namespace)::', '')[:110]}")
print(len(evs), "launches,", sum(e.device_time for e in evs), "us")
