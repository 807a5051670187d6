"""Device-side census of the supernet (config 5) weights pass: which host-side op and which npp_amd source line launched each Fill /
copyBuffer / Memcpy / Memset / ATen kernel in one eager step (torch.profiler, device activities + python stacks)."""
import collections
import os
import sys
from types import SimpleNamespace as NS
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd.model_augment import set_compute_dtype
from npp_amd.model_search_interact import Network as SearchNetwork
from npp_amd.criterion import Criterion_par, Criterion_pose
from npp_amd.optim import FusedAdam
from npp_amd.synth import synth_batch
from npp_amd.train_step import TrainStep

dev = torch.device("cuda:0")
set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
net = SearchNetwork(NS(DATASET=NS(NUM_CLASSES=20, NUM_JOINTS=16), SEARCH=NS(LAYERS=16, INIT_CHANNELS=32),
                       MODEL=NS(DECONV_WITH_BIAS=False, HEAD='PSP', REFINE_LAYERS=1))).to(dev).train()
cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
arch_ids = {id(a) for a in net.arch_parameters()}
opt = FusedAdam([q for q in net.parameters() if id(q) not in arch_ids] + list(cp.parameters()) + list(cq.parameters()), lr=1e-4)
images, lpar, lpose, _ = synth_batch(8, 384, seed=0)
images = torch.from_numpy(images).to(dev)
lpar = [torch.from_numpy(a).to(dev) for a in lpar]
lpose = [torch.from_numpy(a[:, :-1].copy()).to(dev) for a in lpose]
step = TrainStep(net, cp, cq, opt, graph=False)
for _ in range(3):
    step(images, lpar, lpose)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(images, lpar, lpose)
    torch.cuda.synchronize()
dev_cnt = collections.Counter()
dev_us = collections.Counter()
op_cnt = collections.Counter()
for e in prof.events():
    if str(e.device_type).endswith("CUDA"):
        nm = e.name
        if "npp" in nm or "anonymous" in nm:
            nm = "(library kernels)"
        dev_cnt[nm[:90]] += 1
        dev_us[nm[:90]] += e.device_time if hasattr(e, "device_time") else 0
    elif e.kernels:
        for k in e.kernels:
            kn = k.name
            if "anonymous" in kn:
                continue
            fr = [s for s in (e.stack or []) if "npp_amd" in s]
            p = e
            while not fr and p.cpu_parent is not None:
                p = p.cpu_parent
                fr = [s for s in (p.stack or []) if "npp_amd" in s]
            site = " < ".join(s.split("npp_amd/")[-1][:60] for s in fr[:3])
            op_cnt[(kn[:60], e.name[:30], site, str(e.input_shapes)[:40])] += 1
print("device-side events of one eager step:")
for k, n in dev_cnt.most_common(25):
    print(f"{n:6d} {dev_us[k] / 1e3:8.3f} ms  {k}")
print("host op -> non-library device work:")
for k, n in op_cnt.most_common(70):
    print(f"{n:6d} {k}")
