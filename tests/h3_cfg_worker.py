"""Child process of test_kernels_gpu.test_h3_tile_configurations_in_subprocess (NPP_H3_CFG in the environment): the 3x3
halo-footprint parity cases, bf16, through the selected tile configuration of conv_h3_kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import test_kernels_gpu as T

assert os.environ.get("NPP_H3_CFG") in ("1", "3", "4")
n = 0
for case in T.H3_CASES:
    T.test_conv_fwd_bwd(case, torch.bfloat16, 3e-2)
    n += 1
assert n >= 4
print("h3 cfg ok", n)
