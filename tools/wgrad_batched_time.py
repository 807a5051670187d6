"""The batched weight-gradient launch (npp_conv_wgrad_batched) on the 3x3 jobs of one training step, bf16, N = 16: HIP events over 5 launches.
    python3 tools/wgrad_batched_time.py [small]        (env: NPP_WG9=0 the 128 x 128 kernel for every job, NPP_WG9_STAGES=n)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _ops as K
from npp_amd import _lib as L
from npp_amd._lib import lib, check, desc, geom

dev = torch.device("cuda:0")
JOBS = [(128, 128, 96, 30), (384, 128, 96, 2), (256, 256, 48, 6)]
K1 = len(sys.argv) > 1 and sys.argv[1] == "k1"
if K1:      # the 1x1 jobs of a step (k = 1)
    JOBS = [(512, 128, 96, 12), (128, 128, 96, 15), (1024, 512, 96, 2), (1024, 384, 96, 2), (384, 128, 96, 6), (512, 256, 96, 4), (256, 64, 48, 12),
            (512, 128, 24, 14), (64, 64, 48, 14), (512, 512, 24, 7), (256, 256, 48, 6), (1024, 256, 12, 10), (128, 128, 24, 14), (256, 128, 48, 7)]
if len(sys.argv) > 1 and sys.argv[1] == "small":      # the small-map 3x3 jobs
    JOBS = [(128, 128, 24, 35), (256, 256, 12, 32), (512, 512, 24, 2)]
if len(sys.argv) > 1 and sys.argv[1] == "all":
    JOBS += [(128, 128, 24, 35), (256, 256, 12, 32), (512, 512, 24, 2)]
n = 16
items, keep, flops = [], [], 0.0
for cin, cout, hw, cnt in JOBS:
    x = K.cast(torch.randn(n, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
    dy = K.cast(torch.randn(n, cout, hw, hw, device=dev).contiguous(memory_format=torch.channels_last), torch.bfloat16)
    kk = 1 if K1 else 3
    g = geom(kk, kk, 1, 1, kk // 2, kk // 2, 1, 1, 1, 1)
    nel = int(lib().npp_packed_weight_elems(cout, cin, kk, kk, 0))
    nsl = int(lib().npp_conv_wgrad_batched_slabs(C.byref(desc(x)), C.byref(desc(dy)), C.byref(g)))
    for _ in range(cnt):
        acc = torch.zeros(max(nsl, 1) * nel, dtype=torch.float32, device=dev)
        items.append((x, dy, acc, g, nsl))
        flops += 2.0 * n * hw * hw * cin * cout * kk * kk
    keep.append((x, dy))
    print(f"{cin}->{cout} @{hw}: {nsl} slabs x {nel * 4 / 1e6:.2f} MB", flush=True)
m = len(items)
arr = (L.NppWgradItem * m)()
for i, (x, dy, acc, g, nsl) in enumerate(items):
    arr[i].x, arr[i].dy, arr[i].dw_packed, arr[i].g, arr[i].nslabs = desc(x), desc(dy), acc.data_ptr(), g, nsl
nb = int(lib().npp_conv_wgrad_batched_ws(m))
pin = torch.empty(nb, dtype=torch.uint8).pin_memory()
dv = torch.empty(nb, dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    check(lib().npp_conv_wgrad_batched(C.cast(arr, C.c_void_p), m, pin.data_ptr(), dv.data_ptr(), nb, s), "batched")
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
e0.record()
for _ in range(reps):
    check(lib().npp_conv_wgrad_batched(C.cast(arr, C.c_void_p), m, pin.data_ptr(), dv.data_ptr(), nb, s), "batched")
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"{m} jobs, {flops / 1e9:.0f} GFLOP: {ms:.3f} ms = {flops / ms / 1e9:.0f} TF/s  (NPP_WG9={os.environ.get('NPP_WG9', '1')}, stages {os.environ.get('NPP_WG9_STAGES', '64')}, ring {os.environ.get('NPP_WGB_RING', '2')})", flush=True)
