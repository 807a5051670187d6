"""Sanity anchor of the CPU baseline (the oracle port must sit within +-30 % of the real reference on the same
cores).  Runs in the build container only (it imports /root/reference through oracle/make_golden.py):
    python oracle/cpu_anchor.py [threads]
times config 1 (fixed genotype, C = 64, 1 x 3 x 384 x 384, train mode: forward + both criteria + backward, fp32) through the
reference's own modules and through oracle.nppnet_oracle.train_step_loss -- seconds per iteration, median of 3 after 1 warm-up."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.set_num_threads(threads)
from oracle import make_golden as G      # noqa: E402  (imports the reference)
from oracle import nppnet_oracle as O    # noqa: E402
from npp_amd.synth import synth_batch    # noqa: E402


def timed(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


net = G.RefNetwork(G.cfg(64))
G.load_synth(net, 0)
net.train()
_im, _lpar, _lpose, _meta = synth_batch(1, 384, seed=0)
_im = torch.from_numpy(_im)
_lpar = [torch.from_numpy(a) for a in _lpar]
_lpose = [torch.from_numpy(a[:, :-1]) for a in _lpose]
_cpose = G.Criterion_pose(out_len=2, use_target_weight=False)
_cpar = G.Criterion_par(out_len=2)
_tw = torch.from_numpy(_meta["pose_weight"])


def ref_step():      # the loop body of core/function.py:72-107 on the reference's own modules
    pose_list, par_list = net(_im)
    loss = (_cpar(par_list, _lpar).unsqueeze(0) + _cpose(pose_list, _lpose, target_weight=_tw).unsqueeze(0)).mean()
    net.zero_grad()
    loss.backward()


t_ref = timed(ref_step)
tensors = {k: v.detach().clone() for k, v in net.state_dict().items()}
for k, v in tensors.items():
    if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
        v.requires_grad_(True)
images, lpar, lpose, _ = synth_batch(1, 384, seed=0)
images = torch.from_numpy(images)
lpar = [torch.from_numpy(a) for a in lpar]
lpose = [torch.from_numpy(a[:, :-1].copy()) for a in lpose]
lam_pose = torch.full((2,), -2.5, requires_grad=True)
lam_par = torch.full((2,), 2.3, requires_grad=True)


def one():
    for v in tensors.values():
        v.grad = None
    loss, _, _, _ = O.train_step_loss(tensors, images, lpar, lpose, lam_pose, lam_par)
    loss.backward()


t_or = timed(one)
print(f"threads {threads}: reference {t_ref:.2f} s/iter, "
      f"oracle {t_or:.2f} s/iter, ratio {t_or / t_ref:.2f}")
