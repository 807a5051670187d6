"""What is left for the fan-out nodes to add after the conv epilogues accumulate (one eager training step, batch 16, bf16):
every npp_add_n call by tensor shape, operand count and which kernels produced the operands."""
import os, sys, collections
os.environ["NPP_STREAMS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from npp_amd import _ops as K
from npp_amd.model_augment import Network, set_compute_dtype
from npp_amd.criterion import Criterion_par, Criterion_pose
from npp_amd.synth import synth_batch
dev = torch.device("cuda:0")
set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
net = Network(bench.cfg_ns()).to(dev).train()
cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
images, lpar, lpose, _ = synth_batch(16, 384, seed=0)
images = torch.from_numpy(images).to(dev)
lpar = [torch.from_numpy(a).to(dev) for a in lpar]
lpose = [torch.from_numpy(a[:, :-1].copy()).to(dev) for a in lpose]
log = []
orig = K.add_n
def add_n(ts):
    log.append((tuple(ts[0].shape), len(ts), tuple(sorted(getattr(t, "_npp_src", "?") for t in ts))))
    return orig(ts)
K.add_n = add_n
# tag the gradients the wrappers hand back
def tag(cls, name, idx=0):
    bw = cls.backward
    def wrapped(ctx, *a):
        out = bw(ctx, *a)
        t = out[idx] if isinstance(out, tuple) else out
        if isinstance(t, torch.Tensor):
            try:
                t._npp_src = name
            except Exception:
                pass
        return out
    cls.backward = staticmethod(wrapped)
for cls, name in ((K._Conv2d, "conv"), (K._DwConv2d, "dw"), (K._Pool3x3, "pool3"), (K._Pool2x2, "pool2"), (K._SEScale, "se"),
                  (K._Bilinear, "bilinear"), (K._BnAdd, "bnadd_a"), (K._Concat, "concat"), (K._ConcatAlias, "concat_alias"),
                  (K._FanOut, "fanout")):
    tag(cls, name)
for _ in range(2):
    log.clear()
    pose_list, par_list = net(images)
    loss = (cq(par_list, lpar).unsqueeze(0) + cp(pose_list, lpose).unsqueeze(0)).mean()
    net.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize()
tot = collections.Counter()
byt = collections.Counter()
for shape, n, kinds in log:
    mb = 2 * shape[0] * shape[1] * shape[2] * shape[3] / 1e6
    tot[(shape, n, kinds)] += 1
    byt[(shape, n, kinds)] += (n + 1) * mb
print(f"{len(log)} add_n calls, {sum(byt.values()) / 1e3:.2f} GB;  fan accumulator claims (stored, added, private): {K.FAN_STATS}")
for k, v in sorted(byt.items(), key=lambda kv: -kv[1])[:40]:
    print(f"  {v:8.1f} MB  x{tot[k]:3d}  shape {k[0]}  n={k[1]}  from {k[2]}")
