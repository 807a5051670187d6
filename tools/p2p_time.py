"""Duration of one peer-to-peer exchange kernel on a 1-rank world (csrc/p2p.hip; NPP_P2P_ALONE=1): 200 dependent exchanges of n doubles
replayed from a hipGraph.      NPP_P2P_ALONE=1 python3 tools/p2p_time.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NPP_P2P_ALONE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29688")
import torch
import torch.distributed as dist
from npp_amd import comm

dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
assert comm.enable_p2p(None)
from npp_amd import _lib
print("mailbox allocation kind (0 uncached, 1 fine-grained, 2 plain):", _lib.lib().npp_p2p_alloc_kind(), flush=True)
st = torch.cuda.Stream()
for n in (64, 512, 4096, 32768):
    v = torch.ones(n, dtype=torch.float64, device="cuda")
    slabs = torch.ones(16 * n, dtype=torch.float64, device="cuda")
    for form in ("plain", "slabs"):
        with torch.cuda.stream(st):
            def one():
                if form == "plain":
                    assert comm.p2p_exchange(v, None)
                else:
                    assert comm.p2p_exchange_slabs([(slabs, n, 16, n // 2, (None, None, None, None), True)], None)
            for _ in range(3):
                one()
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(200):
                    one()
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        print(f"n = {n:6d} doubles, {form:6s}: {a.elapsed_time(b) * 1e3 / 200:6.2f} us per exchange", flush=True)
        del g
comm.disable_p2p()
dist.destroy_process_group()
