"""Child process of test_kernels_gpu.test_deterministic_slab_weight_gradient_in_subprocess (NPP_WGRAD_SLABS=1)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import test_kernels_gpu as T
from npp_amd import _ops as K
from npp_amd._lib import lib

assert os.environ.get("NPP_WGRAD_SLABS") == "1"
n = 0
for case in T.CONV_CASES:
    cin, cout, k, stride, pad, dil, H, W, N = case[:9]
    if k == 3 and stride == 1 and dil == 1 and cin % 128 == 0 and cout % 64 == 0 and W % 32 == 0 and N * H * W >= 30000:
        T.test_conv_fwd_bwd(case, torch.bfloat16, 3e-2)
        n += 1
assert n >= 3, n
dev = torch.device("cuda:0")
x = K.cast(torch.randn(8, 128, 64, 64, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
gy = K.cast(torch.randn(8, 128, 64, 64, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
geo = K.geom(3, 3, 1, 1, 1, 1, 1, 1, 1, 1)
assert lib().npp_conv_wgrad_splits(K._byref(x), K._byref(gy), C.byref(geo)) > 0, "the slab kernel did not take the shape"
w = (torch.randn(128, 128, 3, 3, device=dev) * 0.03).requires_grad_(True)
grads = []
for _ in range(2):
    w.grad = None
    xx = x.detach().requires_grad_(True)
    y, _ = K.conv2d(xx, w, None, 1, 1, 1, relu_in=True)
    y.backward(gy)
    torch.cuda.synchronize()
    grads.append(w.grad.clone())
assert torch.equal(grads[0], grads[1]), "slab weight gradients differ between two runs"
print("wgrad slabs ok", n)
