"""Per-family kernel time and algorithmic bandwidth of one eager single-stream training step (HIP events around every launch
of the family): python tools/fam_prof.py"""
import os, sys, ctypes as C
os.environ["NPP_STREAMS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from npp_amd import _lib
from npp_amd.model_augment import Network, set_compute_dtype
from npp_amd.criterion import Criterion_par, Criterion_pose
from npp_amd.synth import synth_batch
dev = torch.device("cuda:0")
set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
net = Network(bench.cfg_ns()).to(dev).train()
cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
images, lpar, lpose, _ = synth_batch(16, 384, seed=0)
images = torch.from_numpy(images).to(dev)
lpar = [torch.from_numpy(a).to(dev) for a in lpar]
lpose = [torch.from_numpy(a[:, :-1].copy()).to(dev) for a in lpose]
def step():
    pose_list, par_list = net(images)
    loss = (cq(par_list, lpar).unsqueeze(0) + cp(pose_list, lpose).unsqueeze(0)).mean()
    net.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize()
step(); step()
L = _lib.lib()
print(f"{'family':>12} {'launches':>8} {'ms':>8} {'GB':>8} {'GB/s':>8} {'TF/s':>8}")
for fam, code in _lib.FAM.items():
    if fam == "none":
        continue
    L.npp_prof_begin(code, _lib.NPP_BF16)
    step()
    ms, fl, by, nl = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
    L.npp_prof_end(C.byref(ms), C.byref(fl), C.byref(by), C.byref(nl))
    if nl.value:
        print(f"{fam:>12} {nl.value:8d} {ms.value:8.2f} {by.value / 1e9:8.2f} {by.value / 1e6 / max(ms.value, 1e-9):8.0f} {fl.value / 1e9 / max(ms.value, 1e-9):8.1f}")
