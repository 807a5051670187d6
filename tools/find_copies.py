import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import cfg_ns
from npp_amd.model_augment import Network, set_compute_dtype
from npp_amd.criterion import Criterion_par, Criterion_pose
from npp_amd.synth import synth_batch
dev = torch.device("cuda:0")
set_compute_dtype(torch.bfloat16)
net = Network(cfg_ns()).to(dev).train()
cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
images, lpar, lpose, _ = synth_batch(4, 384, seed=0)
images = torch.from_numpy(images).to(dev)
lpar = [torch.from_numpy(a).to(dev) for a in lpar]
lpose = [torch.from_numpy(a[:, :-1].copy()).to(dev) for a in lpose]
opt = torch.optim.Adam(list(net.parameters()) + list(cp.parameters()) + list(cq.parameters()), lr=1e-4)
def step():
    pl, pr = net(images)
    loss = (cq(pr, lpar).unsqueeze(0) + cp(pl, lpose).unsqueeze(0)).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_input_shape=True)
rows = [(e.count, e.key, str(e.input_shapes)[:80]) for e in ev if e.key in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::zeros", "aten::to", "aten::_to_copy", "aten::cat", "aten::sum", "aten::mul", "aten::div")]
rows.sort(reverse=True)
for r in rows[:40]:
    print(r)
