"""Which Python lines issue torch (ATen) work during one eager training step: TorchDispatchMode counts aten ops that launch
kernels, keyed by the innermost npp_amd / bench frame."""
import os, sys, collections, traceback
os.environ["NPP_STREAMS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from npp_amd.model_augment import Network, set_compute_dtype
from npp_amd.criterion import Criterion_par, Criterion_pose
from npp_amd.optim import FusedAdam
from npp_amd.synth import synth_batch
dev = torch.device("cuda:0")
set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
if os.environ.get("MODEL") == "search":
    from types import SimpleNamespace as NS
    from npp_amd.model_search_interact import Network as SearchNetwork
    net = SearchNetwork(NS(DATASET=NS(NUM_CLASSES=20, NUM_JOINTS=16), SEARCH=NS(LAYERS=16, INIT_CHANNELS=32),
                           MODEL=NS(DECONV_WITH_BIAS=False, HEAD='PSP', REFINE_LAYERS=1))).to(dev).train()
else:
    net = Network(bench.cfg_ns()).to(dev).train()
cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
_arch = {id(a) for a in net.arch_parameters()} if hasattr(net, "arch_parameters") else set()
opt = FusedAdam([q for q in net.parameters() if id(q) not in _arch] + list(cp.parameters()) + list(cq.parameters()), lr=1e-4)
images, lpar, lpose, _ = synth_batch(8 if os.environ.get("MODEL") == "search" else 16, 384, seed=0)
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
SKIP = ("aten.view", "aten.detach", "aten.alias", "aten.permute", "aten.slice", "aten.empty", "aten.as_strided", "aten.reshape",
        "aten._unsafe_view", "aten.unsqueeze", "aten.select", "aten.expand", "aten.t.", "aten.transpose", "aten.squeeze",
        "aten.empty_strided", "aten.new_empty", "aten._local_scalar_dense", "aten.is_", "aten.stride", "aten.size", "aten.sym_")
cnt = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            fr = "?"
            for f in reversed(traceback.extract_stack()[:-1]):
                if ("npp_amd" in f.filename or "bench" in f.filename or "aten_sites" in f.filename) and "aten_sites.py" not in f.filename:
                    fr = f"{os.path.basename(f.filename)}:{f.lineno}"
                    break
            cnt[(name, fr)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    step()
for (name, fr), n in cnt.most_common(60):
    print(f"{n:5d} {name:34s} {fr}")
