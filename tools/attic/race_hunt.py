"""Race hunt: the same forward + backward of the tiny f32 network N times in one process (same weights, same batch); every
parameter gradient of every iteration is compared with the element-wise median over iterations.  Rounding-order noise (float
atomics, arg-max flips on 2x2 maps) stays below ~1e-2; a missing cross-stream dependency shows as an outlier iteration, and the
parameters listed for it (in module order) say where the wrong values entered.
  python tools/race_hunt.py [iters] [sync]      env: NPP_STREAMS, NPP_SYNCBN_STREAMS ..."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
if os.environ.get("RACE_POISON"):      # tools/poison_alloc.cpp: memory nobody wrote reads as NaN
    torch.cuda.memory.change_current_allocator(torch.cuda.memory.CUDAPluggableAllocator(
        os.path.join(REPO, "tools", "libpoison_alloc.so"), "poison_malloc", "poison_free"))
from helpers import load_golden, synth_tensors, template_from_golden      # noqa: E402
from test_syncbn_gpu import _cfg, _outputs      # noqa: E402
from npp_amd import _ops as K      # noqa: E402
from npp_amd.model_augment import Network, set_compute_dtype      # noqa: E402
from npp_amd.synth import synth_batch      # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sync = len(sys.argv) > 2 and sys.argv[2] == "sync"
dev = torch.device("cuda:0")
if sync:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    dist.init_process_group("nccl", rank=0, world_size=1)
    K._SYNC_EVEN_ALONE = True
g = load_golden("tiny_net.npz")
set_compute_dtype(torch.float32)
net = Network(_cfg(int(g["C"])))
state = synth_tensors(template_from_golden(g), 0)
net.load_state_dict(state)
if sync:
    net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
net = net.to(dev).train()
images, _, _, _ = synth_batch(int(g["n"]), int(g["size"]), seed=0)
x = torch.from_numpy(images).to(dev)
names = [n for n, _ in net.named_parameters()]
params = dict(net.named_parameters())
runs = []
for it in range(iters):
    net.load_state_dict({k: v.to(dev) for k, v in state.items()})       # running statistics back to the start
    pose_list, par_list = net(x)
    loss = sum((o.float() ** 2).sum() for o in _outputs(pose_list, par_list))
    net.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize()
    runs.append({n: params[n].grad.detach().double().cpu().numpy().ravel() for n in names if params[n].grad is not None})
used = [n for n in names if n in runs[0]]
med = {n: np.median(np.stack([r[n] for r in runs]), axis=0) for n in used}
gscale = float(np.median([np.linalg.norm(v) for v in med.values()]))
worst = []
for it, r in enumerate(runs):
    errs = {n: float(np.linalg.norm(r[n] - med[n]) / max(np.linalg.norm(med[n]), 1e-30)) for n in used}
    bad = [(n, e) for n, e in errs.items() if not e <= 3e-2 and (np.linalg.norm(med[n]) > 1e-4 * gscale or e != e)]
    worst.append(max([e for _, e in bad], default=0.0))
    print(f"iter {it:3d}  outlier tensors: {len(bad)}  non-finite: {sum(e != e for e in errs.values())}", flush=True)
    if bad:
        for n, e in bad[:6] + bad[-6:]:
            print(f"      {n:55s} {e:.3e}")
print("SUMMARY outlier iterations", sum(not w <= 3e-2 for w in worst), "of", iters)
