"""How many `ReLU -> conv` data gradients of one training step read a bit-mask (tools: NPP_RELU_BITS)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import cfg_ns
from npp_amd import _ops as K
from npp_amd.model_augment import Network, set_compute_dtype
from npp_amd.criterion import Criterion_par, Criterion_pose
from npp_amd.synth import synth_batch
set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda:0")
net = Network(cfg_ns()).to(dev).train()
cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
images, lpar, lpose, _ = synth_batch(4, 384, seed=0)
K.SHAPE_LOG = []
po, pa = net(torch.from_numpy(images).to(dev))
loss = (cq(pa, [torch.from_numpy(a).to(dev) for a in lpar]).unsqueeze(0) + cp(po, [torch.from_numpy(a[:, :-1].copy()).to(dev) for a in lpose]).unsqueeze(0)).mean()
miss_after_fwd = K.MASK_STATS[2]
loss.backward()
torch.cuda.synchronize()
print("bit-mask dgrads / unsupported kernel / no mask found:", K.MASK_STATS, "forward misses", miss_after_fwd)
