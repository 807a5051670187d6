import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from types import SimpleNamespace as NS
from helpers import load_golden, synth_tensors, template_from_golden, rel_err
from npp_amd.model_search_interact import Network
from npp_amd.model_augment import set_compute_dtype
from npp_amd.synth import synth_batch
g = load_golden("search_net.npz")
dev = torch.device("cuda:0")
set_compute_dtype(torch.float32)
cfg = NS(DATASET=NS(NUM_CLASSES=20, NUM_JOINTS=16), SEARCH=NS(LAYERS=16, INIT_CHANNELS=int(g["C"])),
         MODEL=NS(DECONV_WITH_BIAS=False, HEAD='PSP', REFINE_LAYERS=1))
images = torch.from_numpy(synth_batch(int(g["n"]), int(g["size"]), seed=0)[0]).to(dev)
res = {}
for mode in ("1", "2", "2"):
    os.environ["NPP_STREAMS"] = mode
    net = Network(cfg)
    net.load_state_dict(synth_tensors(template_from_golden(g), 0))
    net = net.to(dev)
    for tr in (False, True):
        net.train(tr)
        if tr:
            p, q = net(images)
            loss = sum((t.float() ** 2).mean() for pair in p + q for t in pair)
            net.zero_grad()
            loss.backward()
        else:
            with torch.no_grad():
                p, q = net(images)
        torch.cuda.synchronize()
        outs = [t.detach().float().cpu().numpy() for pair in p + q for t in pair]
        if tr:
            outs += [a.grad.detach().float().cpu().numpy() for a in net.arch_parameters()]
        key = (mode, tr)
        if ("1", tr) in res and mode != "1":
            print(mode, "train" if tr else "eval", [f"{rel_err(a, b):.1e}" for a, b in zip(outs, res[("1", tr)])])
        res.setdefault(key, outs)
