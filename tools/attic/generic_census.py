"""Which conv shapes still run on the generic kernels (conv_igemm_kernel / conv_wgrad_kernel)?  One eager training step of the
bench model with NPP_TRACE_GENERIC=1 (a stderr line per launch of a generic kernel), counted per shape.
    NPP_TRACE_GENERIC=1 python3 tools/generic_census.py 2>&1 | sort | uniq -c | sort -rn"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NPP_TRACE_GENERIC", "1")
import torch
import bench
from npp_amd.model_augment import Network, set_compute_dtype

dev = torch.device("cuda:0")
torch.manual_seed(0)
set_compute_dtype(torch.bfloat16)
net = Network(bench.cfg_ns()).to(dev).train()
x = torch.randn(16, 3, 384, 384, device=dev)
for it in range(2):
    if it == 1:
        sys.stderr.write("npp-generic ---- step ----\n")
        sys.stderr.flush()
    out = net(x)
    flat = []
    def walk(o):
        if isinstance(o, (list, tuple)):
            for q in o: walk(q)
        elif torch.is_tensor(o): flat.append(o)
    walk(out)
    sum(o.float().sum() for o in flat).backward()
    torch.cuda.synchronize()
