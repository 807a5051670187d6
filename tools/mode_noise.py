"""Noise floor of the tiny network's gradients: mode 1 vs mode 1 (run to run) against mode 2 vs mode 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_ops_gpu as TO
from helpers import load_golden, rel_err
from npp_amd.synth import synth_batch
g = load_golden("tiny_net.npz")
dev = torch.device("cuda:0")
images, _, _, _ = synth_batch(int(g["n"]), int(g["size"]), seed=0)
x = torch.from_numpy(images).to(dev)
def run(mode):
    os.environ["NPP_STREAMS"] = mode
    net = TO._build_net(int(g["C"]), torch.float32, g).train()
    p, q = net(x)
    loss = sum((t.float() ** 2).mean() for pair in p + q for t in pair)
    net.zero_grad(); loss.backward(); torch.cuda.synchronize()
    return {k: v.grad.detach().float().cpu().numpy() for k, v in net.named_parameters() if v.grad is not None}
def top(a, b):
    errs = sorted(((rel_err(runs[b][k], runs[a][k]), k) for k in runs[a] if np.abs(runs[a][k]).max() >= 1e-6), reverse=True)[:2]
    return ", ".join(f"{k} {e:.2e}" for e, k in errs)


seq = os.environ.get("SEQ", "1,1,1,1,1,1").split(",")
runs = {}
for n, m in enumerate(seq):
    runs[f"{n}:{m}"] = run(m)
keys = list(runs)
for b in keys[1:]:
    print(keys[0], "vs", b, ":", top(keys[0], b), flush=True)
