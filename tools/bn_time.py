"""BatchNorm backward, two-sided node (dout -> dx_a, dx_b): one-launch kernel (csrc/bn_one.hip) against reduce2_acc + apply2_fin.
50 dependent repetitions captured in a hipGraph (launch-gap free, as in the training step), best of 3 replays (bf16, N = 16).
    python3 tools/bn_time.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
from npp_amd import _lib as L
from npp_amd._lib import lib, check, desc

dev = torch.device("cuda:0")
R = 16
for c, hw in [(32, 96), (64, 48), (128, 24), (256, 12), (64, 96), (128, 48)]:
    n = 16
    mk = lambda: K.cast(torch.randn(n, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
    dout, a, b, dxa, dxb = mk(), mk(), mk(), mk(), mk()
    mia = torch.cat([torch.zeros(c), torch.ones(c)]).to(dev)
    mib = mia.clone()
    g = torch.ones(c, device=dev)
    dg = [torch.empty(c, device=dev) for _ in range(4)]
    bar = torch.zeros(24 * 257, dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    nb = lib().npp_reduce_blocks(n * hw * hw, c, L.NPP_BF16)
    sums = [torch.zeros(R * 3 * c, dtype=torch.float64, device=dev) for _ in range(60)]
    one_blocks = int(lib().npp_bn_bwd_one_blocks(n * hw * hw, c, L.NPP_BF16, 1))

    def two(i):
        check(lib().npp_bn_bwd_reduce2_acc(C.byref(desc(dout)), C.byref(desc(a)), C.byref(desc(b)), None, mia.data_ptr(), mib.data_ptr(),
                                           sums[i].data_ptr(), nb, s))
        check(lib().npp_bn_bwd_apply2_fin(C.byref(desc(dout)), C.byref(desc(a)), C.byref(desc(b)), None, sums[i].data_ptr(), R,
                                          float(n * hw * hw), mia.data_ptr(), mib.data_ptr(), g.data_ptr(), g.data_ptr(),
                                          dg[0].data_ptr(), dg[1].data_ptr(), dg[2].data_ptr(), dg[3].data_ptr(),
                                          C.byref(desc(dxa)), C.byref(desc(dxb)), s))

    def one(i):
        check(lib().npp_bn_bwd_one2(C.byref(desc(dout)), C.byref(desc(a)), C.byref(desc(b)), sums[i].data_ptr(), float(n * hw * hw),
                                    mia.data_ptr(), mib.data_ptr(), g.data_ptr(), g.data_ptr(), dg[0].data_ptr(), dg[1].data_ptr(),
                                    dg[2].data_ptr(), dg[3].data_ptr(), C.byref(desc(dxa)), C.byref(desc(dxb)), bar.data_ptr(), s))

    res = []
    for fn in (two, one) if one_blocks > 0 else (two,):
        for sm in sums:
            sm.zero_()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            s = st.cuda_stream
            for i in range(5):
                fn(i)
            st.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                s = torch.cuda.current_stream().cuda_stream
                for i in range(50):
                    fn(5 + i)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            for sm in sums:
                sm.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 50)
        res.append(best)
        s = torch.cuda.current_stream().cuda_stream
    print(f"C={c:4d} {hw:3d}^2 N={n}: two launches {res[0]:6.1f} us" + (f"   one launch {res[1]:6.1f} us ({one_blocks} blocks)" if len(res) > 1 else "   (not a shape of the one-launch kernel)"), flush=True)
