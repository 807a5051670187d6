"""Where does the bf16 network leave the f32 one?  Runs the full network (C=64, 384x384) in f32 and bf16 on the GPU with forward
hooks on every cell / stem / layer / head and prints the rel-L2 distance of each hooked output, train and eval mode, N images."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from bench import cfg_ns  # noqa: E402
from npp_amd.model_augment import Network, set_compute_dtype  # noqa: E402
from npp_amd.synth import synth_batch, synth_state_dict  # noqa: E402


def run(dtype, train, n, C=64, size=384):
    set_compute_dtype(dtype)
    torch.manual_seed(0)
    net = Network(cfg_ns(C))
    syn = synth_state_dict(net.state_dict(), 0)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in syn.items()})
    net = net.cuda()
    net.train(train)
    rec = {}

    def hook(name):
        def f(mod, inp, out):
            outs = out if isinstance(out, (tuple, list)) else [out]
            for i, o in enumerate(outs):
                if isinstance(o, torch.Tensor):
                    rec[f"{name}#{i}"] = o.detach().float().cpu().numpy()
        return f
    for name, m in net.named_modules():
        if name.count(".") <= 1 and name and not name[-1:] == "_" and (
                name.startswith(("stem", "cells1.", "cells2.", "upsamples", "pose_net.", "par_net.", "pose_layer", "par_layer",
                                 "edge_layer", "pose_auxlayer", "pose_head.", "par_head.", "edge_head.", "pose_auxnet."))):
            m.register_forward_hook(hook(name))
    images, _, _, _ = synth_batch(n, size, seed=0)
    os.environ["NPP_STREAMS"] = "1"
    with torch.no_grad():
        net(torch.from_numpy(images).cuda())
    torch.cuda.synchronize()
    return rec


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 384
    for train in (False, True):
        a = run(torch.float32, train, n, C, size)
        b = run(torch.bfloat16, train, n, C, size)
        print(f"--- {'train' if train else 'eval'} mode, N={n}, C={C}, {size}x{size}: rel-L2(bf16 - f32) per hooked output")
        for k in a:
            d = np.linalg.norm(a[k].astype(np.float64) - b[k]) / max(np.linalg.norm(a[k].astype(np.float64)), 1e-30)
            print(f"{k:28s} {tuple(a[k].shape)!s:22s} {d:.3e}")


if __name__ == "__main__":
    main()
