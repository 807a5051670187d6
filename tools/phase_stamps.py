"""Phase breakdown of ONE replayed (hipGraph) training step WITHOUT a profiler attached: capturable stamp kernels
(npp_stamp: the GPU's 100 MHz wall clock) at the phase boundaries of Network.forward, of its backward (autograd runs the stamp
nodes on the stream of their forward) and of TrainStep -- rocprofv3 itself changes the overlap of the two branch streams.

    python3 tools/phase_stamps.py [batch] > gpurun_out/stamps.txt
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from npp_amd import _ops as K
from npp_amd.model_augment import Network, set_compute_dtype
from npp_amd.criterion import Criterion_par, Criterion_pose
from npp_amd.optim import FusedAdam
from npp_amd.synth import synth_batch
from npp_amd.train_step import TrainStep

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
net = Network(bench.cfg_ns()).to(dev).train()
cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
opt = FusedAdam(list(net.parameters()) + list(cp.parameters()) + list(cq.parameters()), lr=1e-4)
images, lpar, lpose, _ = synth_batch(batch, 384, seed=0)
images = torch.from_numpy(images).to(dev)
lpar = [torch.from_numpy(a).to(dev) for a in lpar]
lpose = [torch.from_numpy(a[:, :-1].copy()).to(dev) for a in lpose]
K.stamps_begin(dev)
step = TrainStep(net, cp, cq, opt, graph=True, warmup=2)
names = None
for i in range(8):
    if not step.graphed:
        K.STAMPS["names"].clear()          # every eager step (and the capture) re-issues the stamps from index 0
    step(images, lpar, lpose)
torch.cuda.synchronize()
assert step.graphed, "the step was not captured"
import time
t0 = time.perf_counter()
for _ in range(10):
    step(images, lpar, lpose)
torch.cuda.synchronize()
print(f"replayed step: {(time.perf_counter() - t0) * 100:.2f} ms")
buf = K.STAMPS["buf"].cpu().numpy()
names = K.STAMPS["names"]
streams = {}
rows = []
for i, (nm, st) in enumerate(names):
    streams.setdefault(st, len(streams))
    rows.append((int(buf[i]), nm, streams[st]))
t_begin = min(r[0] for r in rows)
rows.sort()
print(f"{len(rows)} stamps on {len(streams)} streams; times in ms from the first stamp (100 MHz clock)")
prev = {}
for t, nm, st in rows:
    ms = (t - t_begin) / 1e5
    d = ms - prev.get(st, 0.0)
    prev[st] = ms
    print(f"  {ms:8.3f}  (+{d:7.3f} on stream {st})  {nm}")
