"""Device-side census: which host-side op (or bare runtime call) launched each Fill / copyBuffer / Memcpy / Memset in one eager TrainStep step
(torch.profiler with device activities; complements tools/aten_sites_prof.py)."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(images, lpar, lpose)
    torch.cuda.synchronize()
dev_cnt = collections.Counter()
op_cnt = collections.Counter()
for e in prof.events():
    if str(e.device_type).endswith("CUDA"):
        nm = e.name
        if "npp" in nm or "anonymous" in nm:
            nm = "(library kernels)"
        dev_cnt[nm[:90]] += 1
    elif e.kernels:
        for k in e.kernels:
            kn = k.name
            if "anonymous" in kn:
                continue
            chain = []
            p = e.cpu_parent
            while p is not None and len(chain) < 3:
                chain.append(p.name[:50])
                p = p.cpu_parent
            op_cnt[(kn[:70], e.name[:40], " <- ".join(chain), str(e.input_shapes)[:50])] += 1
print("device-side events of one eager step:")
for k, n in dev_cnt.most_common(25):
    print(f"{n:6d} {k}")
print("host op -> non-library device work:")
for k, n in op_cnt.most_common(50):
    print(f"{n:6d} {k}")
