"""Host-side cost of one EAGER training step (the unchanged launcher's mode): cProfile over 3 steps, top functions by own time."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from npp_amd.model_augment import Network, set_compute_dtype
from npp_amd.criterion import Criterion_par, Criterion_pose
from npp_amd.optim import FusedAdam
from npp_amd.synth import synth_batch
from npp_amd.train_step import TrainStep
dev = torch.device("cuda:0")
set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
net = Network(bench.cfg_ns()).to(dev).train()
cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
opt = FusedAdam(list(net.parameters()) + list(cp.parameters()) + list(cq.parameters()), lr=1e-4)
images, lpar, lpose, _ = synth_batch(16, 384, seed=0)
images = torch.from_numpy(images).to(dev)
lpar = [torch.from_numpy(a).to(dev) for a in lpar]
lpose = [torch.from_numpy(a[:, :-1].copy()).to(dev) for a in lpose]
step = TrainStep(net, cp, cq, opt, graph=False)
for _ in range(3):
    step(images, lpar, lpose)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step(images, lpar, lpose)
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 3 * 1e3)
# host-only time: how long the Python side needs to ISSUE a step (no sync inside)
t0 = time.perf_counter()
for _ in range(3):
    step(images, lpar, lpose)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host issue ms/step", (t1 - t0) / 3 * 1e3)
pr = cProfile.Profile()
# single-threaded autograd: backward's Python functions run on this thread and show up in the profile
with torch.autograd.set_multithreading_enabled(False):
    step(images, lpar, lpose)
    pr.enable()
    for _ in range(3):
        step(images, lpar, lpose)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
