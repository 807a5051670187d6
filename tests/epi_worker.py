"""Child process of test_kernels_gpu.test_lean_epilogues_equal_the_generic_ones (NPP_EPI_LEAN in the environment, read once per
process): forward with the statistics epilogue and the data gradient through the producer's bit-mask -- first writer and accumulating
second writer of a fan-out tensor -- on shapes of every LDS-DMA conv kernel; raw outputs to an .npz for a bit-exact comparison."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from npp_amd import _ops as K

CASES = [
    # name, cin, cout, k, N, H  (bf16; the kernel each one lands on is named for the reader, the census is not asserted here)
    ("g8_256", 512, 256, 1, 8, 96),        # conv_g8, 256-wide tiles: forward; its data gradient (256 -> 512) too
    ("g8_128", 256, 384, 1, 8, 96),        # conv_g8, BN = 128
    ("g4_pers", 128, 128, 1, 8, 96),       # conv_g4 128 x 128, persistent
    ("g4_64", 128, 128, 3, 4, 24),         # conv_g4 64 x 64, ring of 4
    ("g4_half", 32, 64, 1, 4, 48),         # conv_g4, 32 input channels (two taps per K-tile)
    ("h3", 128, 128, 3, 4, 96),            # conv_h3
    ("c32", 32, 32, 3, 6, 96),             # conv_c32
]


def main(path):
    dev = torch.device("cuda:0")
    out = {}
    g = torch.Generator().manual_seed(5)
    for name, cin, cout, k, n, h in CASES:
        x_cpu = torch.randn(n, cin, h, h, generator=g)
        w1 = (torch.randn(cout, cin, k, k, generator=g) * (0.5 / (cin * k * k) ** 0.5)).to(dev).requires_grad_(True)
        w2 = (torch.randn(cout, cin, k, k, generator=g) * (0.5 / (cin * k * k) ** 0.5)).to(dev).requires_grad_(True)
        gy1 = torch.randn(n, cout, h, h, generator=g)
        gy2 = torch.randn(n, cout, h, h, generator=g)
        K.fan_reset()
        leaf = K.cast(x_cpu.to(dev).contiguous(memory_format=torch.channels_last), torch.bfloat16).detach().requires_grad_(True)
        x = K.bn_add(K.BnSide(leaf))                    # a producer that leaves the ReLU bit-mask of its output
        y1, st1 = K.conv2d(x, w1, None, 1, k // 2, 1, relu_in=True, want_stats=True)
        y2, st2 = K.conv2d(x, w2, None, 1, k // 2, 1, relu_in=True, want_stats=True)      # second consumer: its dgrad ACCUMULATES
        gd = lambda t: K.cast(t.to(dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
        torch.autograd.backward([y1, y2], [gd(gy1), gd(gy2)])
        torch.cuda.synchronize()
        raw = lambda t: t.detach().contiguous().view(torch.int16).cpu().numpy()
        out[name + "_y1"] = raw(y1)
        out[name + "_y2"] = raw(y2)
        out[name + "_dx"] = raw(leaf.grad)
        out[name + "_st1"] = st1.detach().double().view(-1, 2 * cout).sum(0).cpu().numpy()
        out[name + "_st2"] = st2.detach().double().view(-1, 2 * cout).sum(0).cpu().numpy()
        K.fan_reset()
    np.savez(path, **out)
    print("epi ok", len(CASES))


if __name__ == "__main__":
    main(sys.argv[1])
