"""Child process of test_kernels_gpu.test_g8_taps_variant_in_subprocess (NPP_G8_MAXK=3 NPP_DISABLE_G4=1 in the environment):
the large-map 3x3 parity cases, bf16, through conv_g8_kernel's KxK variant."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import test_kernels_gpu as T

assert os.environ.get("NPP_G8_MAXK") == "3" and os.environ.get("NPP_DISABLE_G4") == "1"
n = 0
for case in T.CONV_CASES:
    cin, cout, k, stride, pad, dil, H, W, N = case[:9]
    if k == 3 and stride == 1 and dil == 1 and N * H * W >= 192 * 256:
        T.test_conv_fwd_bwd(case, torch.bfloat16, 3e-2)
        n += 1
assert n >= 2
assert T._g8_launch_count(128, 128, 3, "conv_g8", 160) == 2, "3x3 did not run on conv_g8_kernel"
print("g8 taps ok", n)
