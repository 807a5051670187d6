import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from npp_amd.model_augment import Network
from npp_amd.optim import FusedAdam
import cProfile, pstats
dev = torch.device("cuda:0")
net = Network(bench.cfg_ns()).to(dev)
params = list(net.parameters())
opt = FusedAdam(params, lr=1e-4)
for it in range(4):
    for p in params:
        p.grad = torch.empty_like(p)
    torch.cuda.synchronize()
    t = time.perf_counter()
    if it == 3:
        pr = cProfile.Profile(); pr.enable()
    opt.step()
    if it == 3:
        pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"step {it}: host {1e3*(t1-t):.1f} ms, total {1e3*(time.perf_counter()-t):.1f} ms")
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
