"""GPU box: census of the non-library device work of ONE eager step of the 1-rank rehearsal of the N > 1 path (SyncBatchNorm over the
p2p mailboxes + gradient reducer): which host op / which npp_amd source line launched it (torch.profiler, python stacks)."""
import collections
import os
import sys
os.environ.setdefault("NPP_P2P_ALONE", "1")
os.environ.setdefault("NPP_P2P_SELFTEST", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from npp_amd import _ops as K
from npp_amd.model_augment import Network, set_compute_dtype
from npp_amd.criterion import Criterion_par, Criterion_pose
from npp_amd.ddp import GradReducer, unused_parameter_names
from npp_amd.optim import FusedAdam
from npp_amd.synth import synth_batch
from npp_amd.train_step import TrainStep

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
K._SYNC_EVEN_ALONE = True
net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(Network(bench.cfg_ns())).to(dev).train()
cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
opt = FusedAdam(list(net.parameters()) + list(cp.parameters()) + list(cq.parameters()), lr=1e-4)
reducer = GradReducer(net, skip=unused_parameter_names(net), always_reduce=True, overlap="tail")
images, lpar, lpose, _ = synth_batch(16, 384, seed=0)
images = torch.from_numpy(images).to(dev)
lpar = [torch.from_numpy(a).to(dev) for a in lpar]
lpose = [torch.from_numpy(a[:, :-1].copy()).to(dev) for a in lpose]
step = TrainStep(net, cp, cq, opt, reducer=reducer, graph=False)
for _ in range(3):
    step(images, lpar, lpose)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(images, lpar, lpose)
    torch.cuda.synchronize()
dev_cnt, dev_us, op_cnt = collections.Counter(), collections.Counter(), collections.Counter()
for e in prof.events():
    if str(e.device_type).endswith("CUDA"):
        nm = e.name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        nm = nm.split("<")[0].split("(")[0] if ("npp" in e.name or "anonymous" in e.name) else nm[:80]
        dev_cnt[nm] += 1
        dev_us[nm] += e.device_time
    elif e.kernels:
        for k in e.kernels:
            if "anonymous" in k.name:
                continue
            fr, p = [], e
            while p is not None and not fr:
                fr = [s for s in (p.stack or []) if "npp_amd" in s]
                p = p.cpu_parent
            site = " < ".join(s.split("npp_amd/")[-1][:50] for s in fr[:3])
            op_cnt[(k.name[:50], e.name[:24], site, str(e.input_shapes)[:36])] += 1
print("device-side events of one eager step:")
for k, n in dev_cnt.most_common(70):
    print(f"{n:6d} {dev_us[k] / 1e3:8.3f} ms  {k}")
print("host op -> non-library device work:")
for k, n in op_cnt.most_common(40):
    print(f"{n:6d} {k}")
dist.destroy_process_group()
