import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import test_train_step_gpu as T
from npp_amd import _ops as K
from npp_amd.criterion import Criterion_par, Criterion_pose
from npp_amd.model_augment import Network, set_compute_dtype
dev = torch.device("cuda:0")
set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
net = Network(T._cfg(32)).to(dev).train()
cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
im, lpar, lpose, _w = T._batch(4, 96, 7, dev)
output_pose, output_par = net(im)
loss = (cq(output_par, lpar).unsqueeze(0) + cp(output_pose, lpose).unsqueeze(0)).mean()
K.DEFER_UNPACK, K.DEFER_WGRAD_MAX_PIX = True, 9300
loss.backward(retain_graph=True)
print("queued", len(K._pending_wgrads))
shapes = [(tuple(it[0].shape), tuple(it[1].shape), it[3].kh, it[3].relu_in) for it in K._pending_wgrads]
K.flush_wgrads(); K.flush_unpacks()
K.DEFER_UNPACK, K.DEFER_WGRAD_MAX_PIX = False, 0
torch.cuda.synchronize()
b = {k: p.grad.detach().float().clone() for k, p in net.named_parameters() if p.grad is not None}
net.zero_grad(set_to_none=True)
loss.backward()
torch.cuda.synchronize()
bad = []
for k, p in net.named_parameters():
    if p.grad is None: continue
    den = float(p.grad.float().norm())
    if den > 1e-8:
        e = float((b[k] - p.grad.float()).norm()) / den
        if e > 1e-4: bad.append((k, tuple(p.shape), round(e, 4), float(b[k].norm()), den))
print("bad", len(bad))
for x in bad[:30]: print(x)
from collections import Counter
print(Counter(shapes).most_common(40))
