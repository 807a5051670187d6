"""Who issues the ATen / runtime kernels inside the step?  One EAGER step of TrainStep under torch.profiler (CPU
activities only: the dispatcher's RecordFunction sees every aten op on every thread, the autograd engine's workers included); the
ops that launch a kernel (fill / zero / copy / clone / add ...) are counted by name, input shape and the chain of enclosing events
(autograd node, Python function)."""
import collections
import os
import sys
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
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    step(images, lpar, lpose)
    torch.cuda.synchronize()
LAUNCH = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add_", "aten::add", "aten::mul", "aten::mul_", "aten::div", "aten::div_", "aten::sum",
          "aten::mean", "aten::cat", "aten::neg", "aten::sub", "aten::clone", "aten::index", "aten::where", "aten::ones_like", "aten::zeros_like")
evs = [e for e in prof.events()]
cnt = collections.Counter()
for e in evs:
    if e.name not in LAUNCH:
        continue
    # skip ops nested inside another launching op (zeros -> zero_ -> fill_: count the innermost once)
    if any(c.name in LAUNCH for c in (e.cpu_children or [])):
        continue
    chain = []
    p = e.cpu_parent
    while p is not None and len(chain) < 4:
        chain.append(p.name[:60])
        p = p.cpu_parent
    stack = [s for s in (e.stack or []) if "npp_amd" in s][:2]
    shapes = str(e.input_shapes)[:60] if e.input_shapes else ""
    cnt[(e.name, " <- ".join(chain), " | ".join(os.path.basename(s.split(",")[0]) for s in stack), shapes)] += 1
print(sum(cnt.values()), "kernel-launching aten ops in one eager TrainStep step")
agg = collections.Counter()
for (name, chain, stack, shapes), n in cnt.items():
    agg[(name, chain, stack)] += n
for (name, chain, stack), n in agg.most_common(45):
    ex = [s for (nm, ch, st, s), m in cnt.items() if nm == name and ch == chain and st == stack][:2]
    print(f"{n:5d} {name:14s} {chain[:110]:110s} {stack[:50]:50s} {ex}")
