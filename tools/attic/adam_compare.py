import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from npp_amd.model_augment import Network, set_compute_dtype
from npp_amd.criterion import Criterion_par, Criterion_pose
from npp_amd.synth import synth_batch
from npp_amd.optim import FusedAdam
dev = torch.device("cuda:0")
set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
net = Network(bench.cfg_ns()).to(dev).train()
cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
images, lpar, lpose, _ = synth_batch(4, 384, seed=0)
images = torch.from_numpy(images).to(dev)
lpar = [torch.from_numpy(a).to(dev) for a in lpar]
lpose = [torch.from_numpy(a[:, :-1].copy()).to(dev) for a in lpose]
state0 = copy.deepcopy(net.state_dict())
params = list(net.parameters()) + list(cp.parameters()) + list(cq.parameters())
def run(kind, steps=6):
    net.load_state_dict(state0)
    with torch.no_grad():
        cp.lamda.fill_(-2.5); cq.lamda.fill_(2.3)
    opt = FusedAdam(params, lr=1e-4) if kind == "fused" else torch.optim.Adam(params, lr=1e-4, fused=True)
    losses = []
    for it in range(steps):
        p, q = net(images)
        loss = (cq(q, lpar).unsqueeze(0) + cp(p, lpose).unsqueeze(0)).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses, [p.detach().clone() for p in params]
la, pa = run("torch")
lb, pb = run("fused")
print("torch", [f"{x:.3f}" for x in la])
print("fused", [f"{x:.3f}" for x in lb])
d = max(float((a - b).abs().max()) for a, b in zip(pa, pb))
print("max param diff after 6 steps", d)

# --- one step from identical state and identical gradients: per-parameter deltas
names = [n for n, _ in net.named_parameters()] + ["cp.lamda", "cq.lamda"]
net.load_state_dict(state0)
p, q = net(images)
loss = (cq(q, lpar).unsqueeze(0) + cp(p, lpose).unsqueeze(0)).mean()
for t in params:
    t.grad = None
loss.backward()
grads = [None if t.grad is None else t.grad.detach().clone() for t in params]
base = [t.detach().clone() for t in params]
def one(kind):
    with torch.no_grad():
        for t, b in zip(params, base):
            t.copy_(b)
    for t, g in zip(params, grads):
        t.grad = None if g is None else g.clone()
    opt = FusedAdam(params, lr=1e-4) if kind == "fused" else torch.optim.Adam(params, lr=1e-4, fused=True)
    opt.step()
    torch.cuda.synchronize()
    return [(t.detach() - b) for t, b in zip(params, base)]
da, db = one("torch"), one("fused")
rows = []
for n, a, b, g in zip(names, da, db, grads):
    if g is None:
        continue
    rows.append((float((a - b).abs().max()), n, float(a.abs().max()), float(b.abs().max()), float(g.abs().max()), g.dtype, tuple(g.shape), g.is_contiguous()))
rows.sort(reverse=True)
for r in rows[:8]:
    print(r)
print("n params with |torch delta| == 0:", sum(1 for r in rows if r[2] == 0.0), "of", len(rows))
