"""Census of the torch (ATen) kernels inside the training step TrainStep captures (324 ATen / runtime launches per
replayed step): one EAGER step of the same callable (same stream topology, deferred weight gradients, batched tails) under a
TorchDispatchMode; every aten op that launches work is counted by the innermost npp_amd frame that issued it.
    python3 tools/aten_sites_step.py > gpurun_out/aten_sites_step.txt"""
import collections
import os
import sys
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
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
SKIP = ("aten.view", "aten.detach", "aten.alias", "aten.permute", "aten.slice", "aten.empty", "aten.as_strided", "aten.reshape",
        "aten._unsafe_view", "aten.unsqueeze", "aten.select", "aten.expand", "aten.t.", "aten.transpose", "aten.squeeze",
        "aten.empty_strided", "aten.new_empty", "aten._local_scalar_dense", "aten.is_", "aten.stride", "aten.size", "aten.sym_",
        "aten.set_", "aten.record_stream", "aten.lift_fresh", "aten.unbind", "aten.split", "aten.narrow", "aten.unflatten", "aten.flatten")
cnt = collections.Counter()


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            frs = []
            for f in reversed(traceback.extract_stack()[:-1]):
                if "npp_amd" in f.filename:
                    frs.append(f"{os.path.basename(f.filename)}:{f.lineno}")
                    if len(frs) == 2:
                        break
            shape = ""
            for a in args:
                if isinstance(a, torch.Tensor):
                    shape = f"{tuple(a.shape)} {str(a.dtype).replace('torch.', '')}"
                    break
            cnt[(name, " < ".join(frs) or "?", shape)] += 1
        return func(*args, **(kwargs or {}))


with Mode():
    step(images, lpar, lpose)
torch.cuda.synchronize()
tot = sum(cnt.values())
print(f"{tot} aten ops that launch work in one eager TrainStep step")
by_site = collections.Counter()
for (name, fr, shape), n in cnt.items():
    by_site[(name, fr)] += n
for (name, fr), n in by_site.most_common(50):
    shapes = collections.Counter({s: m for (nm, f2, s), m in cnt.items() if nm == name and f2 == fr}).most_common(3)
    print(f"{n:5d} {name:30s} {fr:60s} {shapes}")
