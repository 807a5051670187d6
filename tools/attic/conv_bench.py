"""Per-shape conv micro-benchmark on the GPU: fwd / dgrad / wgrad TFLOP/s through the C ABI."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K

SHAPES = [  # cin, cout, k, H, stride
    (128, 128, 3, 96, 1), (1024, 512, 1, 96, 1), (1024, 384, 1, 96, 1), (512, 128, 1, 96, 1), (512, 256, 1, 96, 1),
    (384, 128, 3, 96, 1), (128, 128, 1, 96, 1), (384, 128, 1, 96, 1), (32, 32, 3, 96, 1), (64, 64, 3, 48, 1),
    (128, 128, 3, 24, 1), (256, 256, 3, 12, 1), (256, 256, 3, 48, 1), (1024, 1024, 3, 12, 1), (128, 32, 1, 96, 1),
    (64, 128, 3, 192, 2),
]


def bench(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
    dev = torch.device("cuda:0")
    print(f"N={N} dtype={dt}")
    print(f"{'shape':>28} {'GF':>8} | {'fwd us':>8} {'TF/s':>7} | {'dgrad us':>8} {'TF/s':>7} | {'wgrad us':>8} {'TF/s':>7}")
    for cin, cout, k, H, s in SHAPES:
        x = K.cast(torch.randn(N, cin, H, H, device=dev).contiguous(memory_format=torch.channels_last), dt).requires_grad_(True)
        w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).requires_grad_(True)
        y, _ = K.conv2d(x, w, None, s, k // 2, 1, relu_in=True, want_stats=True)
        gy = K.cast(torch.randn_like(y.float()), dt)
        gf = 2.0 * y.numel() / cout * cout * cin * k * k / 1e9
        t_f = bench(lambda: K.conv2d(x.detach(), w.detach(), None, s, k // 2, 1, relu_in=True, want_stats=True))
        xd = x.detach().requires_grad_(True)
        def dgrad():
            yy, _ = K.conv2d(xd, w.detach(), None, s, k // 2, 1, relu_in=True, want_stats=False)
            torch.autograd.grad(yy, xd, gy)
        def wgrad():
            yy, _ = K.conv2d(x.detach(), w, None, s, k // 2, 1, relu_in=True, want_stats=False)
            torch.autograd.grad(yy, w, gy)
        t_fn = bench(lambda: K.conv2d(x.detach(), w.detach(), None, s, k // 2, 1, relu_in=True, want_stats=False))
        t_d = bench(dgrad) - t_fn
        t_w = bench(wgrad) - t_fn
        print(f"{cin:5d}->{cout:4d} k{k} {H:3d}^2 s{s} {'':>6} {gf:8.2f} | {t_f:8.1f} {gf/t_f*1e3:7.1f} | {t_d:8.1f} {gf/max(t_d,1e-3)*1e3:7.1f} | "
              f"{t_w:8.1f} {gf/max(t_w,1e-3)*1e3:7.1f}")


if __name__ == "__main__":
    main()
