"""Deferred / batched weight gradients written into a GradReducer's bucket slots (overlap=False) vs a plain backward: one forward,
two backward passes (1-rank RCCL group)."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29677")
import torch.distributed as dist
import test_train_step_gpu as T
from npp_amd import _ops as K
from npp_amd.criterion import Criterion_par, Criterion_pose
from npp_amd.ddp import GradReducer, unused_parameter_names
from npp_amd.model_augment import Network, set_compute_dtype
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
net = Network(T._cfg(32)).to(dev).train()
cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
MODE = os.environ.get("NPP_CHECK_REDUCER_MODE", "late")      # late: overlap=False; tail: overlap="tail" with the two-group tail
red = GradReducer(net, skip=unused_parameter_names(net), always_reduce=True, bucket_mb=4, overlap="tail" if MODE == "tail" else False)
if MODE == "tail":
    kinds = [b.kind for b in red.buckets]
    assert "K" in kinds and "O" in kinds and kinds == sorted(kinds), kinds      # the KxK buckets first, never mixed
    from npp_amd.ddp import _is_kxk_weight
    assert all(_is_kxk_weight(p) == (b.kind == "K") for b in red.buckets for p in b.params)
im, lpar, lpose, _w = T._batch(4, 96, 7, dev)
output_pose, output_par = net(im)
loss = (cq(output_par, lpar).unsqueeze(0) + cp(output_pose, lpose).unsqueeze(0)).mean()
net.zero_grad(set_to_none=True)
red.begin_step()
K.DEFER_UNPACK, K.DEFER_WGRAD_MAX_PIX = True, 150000
loss.backward(retain_graph=True)
print("queued", len(K._pending_wgrads), len(K._pending_dw_wgrads), len(K._pending_unpacks))
if MODE == "tail":
    K.flush_wgrads(group="K"); K.flush_unpacks(group="K")
    assert all(it[3].kh * it[3].kw == 1 for it in K._pending_wgrads) and all(it[4] == 1 for it in K._pending_unpacks)
    red.launch_kind("K")
    assert all(b.launched == (b.kind == "K") for b in red.buckets)
    K.flush_wgrads(); K.flush_unpacks()
else:
    K.flush_wgrads(); K.flush_unpacks()
K.DEFER_UNPACK, K.DEFER_WGRAD_MAX_PIX = False, 0
red.finish()
torch.cuda.synchronize()
a = {k: p.grad.detach().float().clone() for k, p in net.named_parameters() if p.grad is not None}
inside = sum(1 for b in red.buckets for p in b.params if p.grad is not None and b.flat.data_ptr() <= p.grad.data_ptr() < b.flat.data_ptr() + b.flat.numel() * 4)
red.remove()
net.zero_grad(set_to_none=True)
loss.backward()
torch.cuda.synchronize()
bad, n = [], 0
for k, p in net.named_parameters():
    if p.grad is None or p.dim() != 4: continue
    den = float(p.grad.float().norm())
    if den > 1e-8:
        n += 1
        e = float((a[k] - p.grad.float()).norm()) / den
        if e > 2e-2: bad.append((k, round(e, 4)))
print("grads inside buckets", inside, "compared", n, "bad", len(bad), bad[:8])
print("REDUCER_DEFER_OK" if not bad and n > 100 and inside > 100 else "REDUCER_DEFER_FAIL")
dist.destroy_process_group()
