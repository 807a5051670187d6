"""auto_graph capture probe: tiny network, the unchanged-launcher loop; PROBE_TWIN=1 interleaves an eager twin network."""
import os
import sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from npp_amd import auto_graph
auto_graph.ENABLED = True
import test_auto_graph_gpu as A
import test_train_step_gpu as T
dev = torch.device("cuda:0")
if len(sys.argv) > 1 and sys.argv[1] == "bf16":
    pass
net, cp, cq = A._setup(dev)
if len(sys.argv) > 1 and sys.argv[1] == "bf16":
    from npp_amd.model_augment import set_compute_dtype
    set_compute_dtype(torch.bfloat16)
twin = None
if os.environ.get("PROBE_TWIN"):
    twin = A._setup(dev)
    twin[0]._auto_graph_off = True
batch = T._batch(int(os.environ.get("PROBE_BATCH", "2")), int(os.environ.get("PROBE_SIZE", "64")), 3, dev)
if os.environ.get("PROBE_SETDEV"):
    torch.cuda.set_device(0)
if os.environ.get("PROBE_TS"):
    from npp_amd.optim import FusedAdam
    from npp_amd.train_step import TrainStep
    ts = TrainStep(net, cp, cq, FusedAdam(list(net.parameters()), lr=1e-4), graph=True, warmup=2)
    net._auto_graph_off = False
if os.environ.get("PROBE_FA"):
    from npp_amd.optim import FusedAdam
    fa = FusedAdam(list(net.parameters()), lr=1e-4)
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
late_zero = bool(os.environ.get("PROBE_LATE_ZERO"))
for it in range(5):
    if not late_zero:
        opt.zero_grad()
    loss = A._loss(net, cp, cq, batch)
    if late_zero:
        opt.zero_grad()
    loss.backward()
    opt.step()
    if twin is not None:
        lt = A._loss(*twin, batch)
        lt.backward()
    if not os.environ.get("PROBE_NO_SYNC"):
        torch.cuda.synchronize()
    print("step", it, float(loss.detach()), "graphed" if net._auto.graph is not None else "eager", flush=True)
print("AUTO_OK")
