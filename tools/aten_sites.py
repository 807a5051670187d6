"""Which Python lines issue torch (ATen) kernels during one eager training step: counts of aten::fill_/zero_/copy_/add/... by
caller frame (torch.profiler with_stack)."""
import os, sys, collections
os.environ["NPP_STREAMS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from npp_amd.model_augment import Network, set_compute_dtype
from npp_amd.criterion import Criterion_par, Criterion_pose
from npp_amd.optim import FusedAdam
from npp_amd.synth import synth_batch
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
def step():
    pose_list, par_list = net(images)
    loss = (cq(par_list, lpar).unsqueeze(0) + cp(pose_list, lpose).unsqueeze(0)).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
step(); step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::sum", "aten::cat", "aten::mul", "aten::div",
                   "aten::empty", "aten::zeros", "aten::to", "aten::_to_copy", "aten::clone", "aten::contiguous"):
        frames = [f for f in (ev.stack or []) if "npp_amd" in f or "bench" in f or "tools" in f]
        cnt[(ev.name, frames[0] if frames else "?")] += 1
for (name, fr), n in cnt.most_common(40):
    print(f"{n:5d} {name:16s} {fr}")
