"""GPU box: how many mailbox exchanges one step of the 1-rank rehearsal issues in forward / backward, and how many segments each carries."""
import collections
import os
import sys
os.environ.setdefault("NPP_P2P_ALONE", "1")
os.environ.setdefault("NPP_P2P_SELFTEST", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29518")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from npp_amd import _ops as K, comm
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
for _ in range(2):
    step(images, lpar, lpose)
torch.cuda.synchronize()
cnt = collections.Counter()
orig_s, orig_p = comm.p2p_exchange_slabs, getattr(comm, "p2p_exchange", None)


import traceback
sites = collections.Counter()


def _site():
    fr = [f for f in traceback.extract_stack()[:-2] if "npp_amd" in f.filename]
    return " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in fr[-4:][::-1])


def slabs(segs, group=None):
    ok = orig_s(segs, group)
    if ok:
        cnt[("fwd" if torch.is_grad_enabled() else "bwd", "slabs", len(segs), sum(s[1] for s in segs))] += 1
        sites[_site()] += 1
    return ok


comm.p2p_exchange_slabs = slabs
if orig_p is not None:
    def plain(t, group=None):
        ok = orig_p(t, group)
        if ok:
            cnt[("fwd" if torch.is_grad_enabled() else "bwd", "plain", 1, t.numel())] += 1
            sites[_site()] += 1
        return ok
    comm.p2p_exchange = plain
step(images, lpar, lpose)
torch.cuda.synchronize()
tot = collections.Counter()
for (ph, kind, nseg, n), c in sorted(cnt.items()):
    tot[(ph, kind)] += c
    print(f"{ph} {kind:5s} segs {nseg} doubles {n:6d}: {c}")
for k, v in sites.most_common(20):
    print(f"{v:4d}  {k}")
print("folded fwd / bwd", K.FOLD_STATS)
print(dict(tot), "multi stats", K.MULTI_STATS, "bn sums", K.BN_SUMS_STATS)
dist.destroy_process_group()
