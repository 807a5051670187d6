"""Which modules still need a stand-alone npp_channel_stats pass (statistics not produced by the producing kernel)?"""
import os, sys, collections, traceback, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from npp_amd import _ops as K
from npp_amd.model_augment import Network, set_compute_dtype
dev = torch.device("cuda:0")
set_compute_dtype(torch.bfloat16)
net = Network(bench.cfg_ns()).to(dev).train()
orig = K.channel_stats
sites = collections.Counter()
def counted(x, level=1):
    fr = [f for f in traceback.extract_stack() if "npp_amd" in f.filename and "_ops.py" not in f.filename]
    sites[(fr[-1].filename.split("/")[-1], fr[-1].lineno, fr[-1].name, tuple(x.shape))] += 1
    return orig(x, level)
K.channel_stats = counted
x = torch.randn(16, 3, 384, 384, device=dev)
net(x)
for k, v in sites.most_common(20): print(v, k)
print("total", sum(sites.values()))
