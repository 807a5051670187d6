"""Uninitialised-read probe: the tiny training step on the poisoning allocator (tools/poison_alloc.cpp).  Any NaN in the loss,
the gradients or the updated parameters means some kernel consumed memory that no kernel had written.
  NPP_SYNC_LAUNCH=1 NPP_TRACE_LAUNCH=1 python tools/poison_probe.py [bf16] 2> trace.log"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
so = os.path.join(REPO, "tools", "libpoison_alloc.so")
torch.cuda.memory.change_current_allocator(torch.cuda.memory.CUDAPluggableAllocator(so, "poison_malloc", "poison_free"))
import test_train_step_gpu as T      # noqa: E402
from npp_amd.model_augment import set_compute_dtype      # noqa: E402

dev = torch.device("cuda:0")
mode = os.environ.get("PROBE_MODE", "tiny")      # tiny | full (C=64, 384^2, batch 1) | search | syncbn (1-rank nccl group)
bf16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"
if mode == "search":
    from npp_amd.optim import FusedAdam
    from npp_amd.train_step import SearchStep
    net, cp, cq, weights = T._search_setup(dev)
    opt = FusedAdam(weights, lr=1e-3)
    sstep = SearchStep(net, cp, cq, opt, FusedAdam(net.arch_parameters(), lr=3e-3, betas=(0.5, 0.999), weight_decay=0.001),
                       graph=False)
    if bf16:
        set_compute_dtype(torch.bfloat16)
    b1, b2 = T._batch(2, 64, 3, dev), T._batch(2, 64, 4, dev)

    def step(*_a):
        l1, l2 = sstep(b1[:3], b2[:3], entropy=True)
        return l1 + l2
    im = lpar = lpose = None
elif mode == "syncbn":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29671")
    dist.init_process_group("nccl", rank=0, world_size=1)
    from npp_amd import _ops as K0
    from npp_amd.criterion import Criterion_par, Criterion_pose
    from npp_amd.ddp import GradReducer, unused_parameter_names
    from npp_amd.model_augment import Network
    from npp_amd.optim import FusedAdam
    from npp_amd.train_step import TrainStep
    K0._SYNC_EVEN_ALONE = True
    set_compute_dtype(torch.bfloat16 if bf16 else torch.float32)
    torch.manual_seed(0)
    net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(Network(T._cfg(8))).to(dev).train()
    cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
    opt = FusedAdam(list(net.parameters()) + list(cp.parameters()) + list(cq.parameters()), lr=1e-3)
    red = GradReducer(net, skip=unused_parameter_names(net), always_reduce=True, bucket_mb=0.5)
    step = TrainStep(net, cp, cq, opt, reducer=red, graph=False, warmup=1)
    im, lpar, lpose, w = T._batch(2, int(os.environ.get("PROBE_SIZE", "64")), 3, dev)
else:
    if mode == "full":
        T._cfg_small = T._cfg
        T._cfg = lambda _c: T._cfg_small(64)
    net, opt, step = T._make(dev, graph=False)
    if bf16:
        set_compute_dtype(torch.bfloat16)
    size = int(os.environ.get("PROBE_SIZE", "384" if mode == "full" else "64"))
    im, lpar, lpose, w = T._batch(1 if mode == "full" else 2, size, 3, dev)
steps = int(os.environ.get("PROBE_STEPS", "3"))
import gc      # noqa: E402
_gc = os.environ.get("PROBE_GC", "")
if _gc == "off":
    gc.disable()
elif _gc.startswith("thr"):
    gc.set_threshold(int(_gc[3:]))
if os.environ.get("PROBE_ANOMALY") or os.environ.get("PROBE_ZERO"):
    from npp_amd import _ops as K
    _orig = K._unpack_or_defer

    def _zeroed(src, dst, *a):      # a deferred weight gradient is unwritten until the flush: not what this probe looks for
        dst.zero_()
        return _orig(src, dst, *a)
    K._unpack_or_defer = _zeroed
if os.environ.get("PROBE_ANOMALY"):
    torch.autograd.set_detect_anomaly(True, check_nan=True)
if os.environ.get("PROBE_SYNC"):
    from npp_amd import _ops as K3
    _flush3 = K3.flush_unpacks
    _mode = os.environ["PROBE_SYNC"]

    def _flush_sync():
        if "before" in _mode:
            torch.cuda.synchronize()
        _flush3()
        if "after" in _mode:
            torch.cuda.synchronize()
    K3.flush_unpacks = _flush_sync
if os.environ.get("PROBE_CPU"):
    # checks that allocate nothing on the device (the address pattern of the failing run stays as it is)
    import numpy as np
    from npp_amd import _ops as K4
    _flush4 = K4.flush_unpacks

    def _nonfinite():
        torch.cuda.synchronize()
        return {n for n, p in net.named_parameters() if p.grad is not None and not np.isfinite(p.grad.cpu().numpy()).all()}

    _var = os.environ.get("PROBE_CPU", "")

    def _flush_var():
        if "touch" in _var:
            n_ = sum(1 for _, p in net.named_parameters() if p.grad is not None)
        if "pre" in _var:
            print("   pre-flush non-finite", len(_nonfinite()))
        _flush4()
        if "post" in _var:
            print("   post-flush non-finite", len(_nonfinite()))

    def _flush_cpu():
        pend = {it[1].data_ptr() for it in K4._pending_unpacks}
        named = list(net.named_parameters())
        deferred = {n for n, p in named if p.grad is not None and p.grad.data_ptr() in pend}
        ptrs = [p.grad.data_ptr() for n, p in named if p.grad is not None]
        before = _nonfinite()
        print("   before flush: non-finite", len(before), "deferred", len(deferred), "pending", len(pend),
              "non-finite and not deferred", sorted(before - deferred)[:10], "distinct grad addresses", len(set(ptrs)), "of", len(ptrs))
        _flush4()
        after = _nonfinite()
        print("   after flush: non-finite", len(after), sorted(after)[:6], flush=True)
    K4.flush_unpacks = _flush_cpu if _var == "1" else _flush_var
if os.environ.get("PROBE_DETAIL"):
    from npp_amd import _ops as K2
    _flush = K2.flush_unpacks

    def _flush_logged():
        torch.cuda.synchronize()
        pend = {it[1].data_ptr() for it in K2._pending_unpacks}
        named = list(net.named_parameters())
        before = [n for n, p in named if p.grad is not None and not torch.isfinite(p.grad).all()]
        deferred = [n for n, p in named if p.grad is not None and p.grad.data_ptr() in pend]
        print("   before flush: non-finite", len(before), "of which deferred", len([n for n in before if n in deferred]),
              "pending", len(pend), "params whose grad is a pending destination", len(deferred))
        print("   non-finite, not deferred:", [n for n in before if n not in deferred][:12])
        _flush()
        torch.cuda.synchronize()
        after = [n for n, p in named if p.grad is not None and not torch.isfinite(p.grad).all()]
        print("   after flush: non-finite", len(after), after[:6], flush=True)
    K2.flush_unpacks = _flush_logged
_col = os.environ.get("PROBE_COLLECT", "")
if _col:
    from npp_amd import _ops as K5
    _flush5 = K5.flush_unpacks

    def _flush_col():
        if "c" in _col:
            gc.collect()
        _flush5()
        if "e" in _col:
            gc.collect()
    K5.flush_unpacks = _flush_col
    pd5 = dict(net.named_parameters())
    if "b" in _col:
        pd5["cells1.8._ops.0.net.1.weight"].register_hook(lambda g: (gc.collect(), None)[1])
    if "B" in _col:
        pd5["cells1.2._ops.0.net.1.weight"].register_hook(lambda g: (gc.collect(), None)[1])
    if "a" in _col:
        net.register_forward_hook(lambda m, i, o: (gc.collect(), None)[1])
for i in range(steps):
    pre = [n for n, p in net.named_parameters() if not torch.isfinite(p).all()]
    if pre:
        print("   params non-finite BEFORE step", i, len(pre), pre[:4])
    loss = step(im, lpar, lpose)
    torch.cuda.synchronize()
    bad_g = [n for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    bad_p = [n for n, p in net.named_parameters() if not torch.isfinite(p).all()]
    bad_b = [n for n, b in net.named_buffers() if b.is_floating_point() and not torch.isfinite(b).all()]
    print("step", i, "loss", float(loss.detach()), "non-finite grads", len(bad_g), bad_g[:8], "params", len(bad_p), bad_p[:4],
          "buffers", len(bad_b), bad_b[:4], flush=True)
    if bad_g and os.environ.get("PROBE_ELEMS"):
        pd = dict(net.named_parameters())
        for n in bad_g[:40]:
            g_ = pd[n].grad
            print(f"     {n:45s} nan elems {int((~torch.isfinite(g_)).sum())} of {g_.numel()}   param nan {int((~torch.isfinite(pd[n])).sum())}")
    if bad_g and os.environ.get("PROBE_LIST"):
        allp = [n for n, p in net.named_parameters() if p.grad is not None]
        print("  finite:", [n for n in allp if n not in bad_g])
        break
print("PROBE_DONE")
