"""Probe: which part of the step breaks hipGraph capture when helper streams are on (tiny net)."""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from npp_amd import _ops as K
from npp_amd.model_augment import Network, set_compute_dtype
from npp_amd.synth import synth_batch

mode = sys.argv[1]
dev = torch.device("cuda:0")
set_compute_dtype(torch.bfloat16)
net = Network(bench.cfg_ns(16)).to(dev).train()
images = torch.from_numpy(synth_batch(2, 128, seed=0)[0]).to(dev)


def fwd():
    with torch.no_grad():
        p, q = net(images)
    return p[1][0].float().sum() + q[1][0].float().sum()


def fwdbwd():
    p, q = net(images)
    loss = sum((t.float() ** 2).mean() for pair in p + q for t in pair)
    net.zero_grad(set_to_none=True)
    loss.backward()
    return loss


fn = fwd if mode == "fwd" else fwdbwd
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        fn()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
K.reset_pools()
g = torch.cuda.CUDAGraph()
print("capturing", mode, flush=True)
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    out = fn()
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print("replayed", float(out))
