"""What brings HIP back after a hipGraph capture that was invalidated half way (an illegal call inside the capture)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npp_amd import _lib
dev = torch.device("cuda:0")
x = torch.ones(1024, device=dev)
side = torch.cuda.Stream()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        y = x * 2
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            z = x + 1
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.current_stream().synchronize()      # illegal inside a capture -> invalidates it
except Exception as e:      # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:80], flush=True)
def attempt(tag, fn):
    try:
        fn()
        print(tag, "-> ok", flush=True)
    except Exception as e:      # noqa: BLE001
        print(tag, "-> FAIL", str(e)[:90].replace("\n", " "), flush=True)
L = _lib.lib()
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
def status(st):
    v = ctypes.c_int(-1)
    rc = hip.hipStreamIsCapturing(ctypes.c_void_p(st), ctypes.byref(v))
    return rc, v.value
cap = torch.cuda.graph.default_capture_stream
print("capture stream status", status(cap.cuda_stream) if cap is not None else None, "side", status(side.cuda_stream),
      "current", status(torch.cuda.current_stream().cuda_stream), flush=True)
if os.environ.get("TRY_END"):
    gr = ctypes.c_void_p()
    for st in ([cap.cuda_stream] if cap is not None else []) + [side.cuda_stream]:
        print("hipStreamEndCapture ->", hip.hipStreamEndCapture(ctypes.c_void_p(st), ctypes.byref(gr)), flush=True)
    print("after end: capture stream", status(cap.cuda_stream), "side", status(side.cuda_stream), flush=True)
if os.environ.get("TRY_SWITCH"):
    torch.cuda.set_stream(torch.cuda.default_stream())
    print("switched to the default stream: status", status(torch.cuda.current_stream().cuda_stream), flush=True)
    side = torch.cuda.Stream()
    print("new side stream status", status(side.cuda_stream), flush=True)
attempt("eager op on current stream", lambda: (x * 3).sum().item())
print("clear ->", L.npp_clear_hip_error(), L.npp_clear_hip_error(), flush=True)
attempt("eager op after clear", lambda: (x * 3).sum().item())
attempt("synchronize", torch.cuda.synchronize)
def on_side():
    with torch.cuda.stream(side):
        v = (x + 5).sum()
    torch.cuda.synchronize()
    return v.item()
attempt("op on the side stream that was in the capture", on_side)
def npp_launch():
    from npp_amd import _ops as K
    a = torch.randn(2, 8, 4, 4, device=dev).contiguous(memory_format=torch.channels_last)
    K.add_n([a, a]); torch.cuda.synchronize()
attempt("npp launch", npp_launch)
