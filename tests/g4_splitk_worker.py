"""Child process of test_kernels_gpu.test_g4_split_k_in_subprocess (NPP_G4_SPLITK and NPP_G4_SPLIT_DBG=1 in the environment): conv_g4's
split-K form (several workgroups per output tile, the last one to arrive sums the partial accumulators and runs the epilogue) on grids
that leave CUs idle -- forward with ReLU + statistics (+ bias, ragged last tile), data gradient through the mask, weight gradient --
against the f32 torch-CPU conv on the same bf16-rounded operands, and twice: the sum is taken in split order whoever arrives last, so
two runs must agree to the last bit."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import test_kernels_gpu as T
from npp_amd import _ops as K

assert os.environ.get("NPP_G4_SPLIT_DBG") == "1"
CASES = [
    # cin, cout, k, stride, pad, dil, H, W, N, relu, bias, slice_pad
    (256, 256, 3, 1, 1, 1, 12, 12, 16, True, False, 0),     # the benched starved shape: 144 tiles x 36 K-tiles
    (128, 128, 3, 1, 1, 1, 24, 24, 2, True, False, 0),      # 36 tiles x 18 K-tiles
    (1024, 256, 1, 1, 0, 1, 12, 12, 16, True, False, 0),    # 1x1, 16 K-tiles
    (256, 256, 3, 1, 1, 1, 11, 13, 3, True, True, 8),       # ragged last tile (generic epilogue), bias, channel-slice input
    (192, 64, 3, 1, 1, 1, 20, 20, 2, False, False, 0),      # three chunks: a share boundary in the middle of a tap
]
for case in CASES:
    T.test_conv_fwd_bwd(case, torch.bfloat16, 3e-2)
# determinism of the combine
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
x = K.cast(torch.randn(16, 256, 12, 12, generator=g).to(dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(dev)
outs = []
for _ in range(3):
    y, st = K.conv2d(x, w, None, 1, 1, 1, relu_in=True, want_stats=True)
    torch.cuda.synchronize()
    outs.append((y.detach().clone(), st.detach().clone()))
for y, st in outs[1:]:
    assert torch.equal(y, outs[0][0]), "split-K outputs differ between runs"
print("g4 splitk ok", len(CASES))
