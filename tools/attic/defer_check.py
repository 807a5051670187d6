import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_train_step_gpu import _cfg, _batch
from npp_amd import _ops as K
from npp_amd.model_augment import Network, set_compute_dtype
from npp_amd.criterion import Criterion_par, Criterion_pose
dev = torch.device("cuda:0")
set_compute_dtype(torch.float32)
torch.manual_seed(0)
net = Network(_cfg(8)).to(dev).train()
cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
im, lpar, lpose, _ = _batch(2, 128, 0, dev)
res = []
for defer in (False, True, True):
    net.zero_grad(set_to_none=True)
    po, pa = net(im)
    loss = (cq(pa, lpar).unsqueeze(0) + cp(po, lpose).unsqueeze(0)).mean()
    K.DEFER_UNPACK = defer
    loss.backward()
    K.DEFER_UNPACK = False
    print("pending", len(K._pending_unpacks))
    K.flush_unpacks()
    torch.cuda.synchronize()
    res.append({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
for j in (1, 2):
    bad = [(k, float((res[0][k] - res[j][k]).abs().max()), float(res[0][k].abs().max())) for k in res[0]
           if float((res[0][k] - res[j][k]).abs().max()) > 1e-2 * float(res[0][k].abs().max()) + 1e-9]
    print("run", j, "bad tensors", len(bad), bad[:5])
