"""Over-read probe: the tiny f32 training step with the caching allocator off (every tensor its own hipMalloc), every launch
traced and synchronised -- a kernel that reads past the end of a tensor is then likely to hit unmapped memory, and the last
traced launch names it.  Usage (GPU box):
  PYTORCH_NO_CUDA_MEMORY_CACHING=1 NPP_SYNC_LAUNCH=1 NPP_TRACE_LAUNCH=1 python tools/overread_probe.py [bf16] 2> trace.log"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import test_train_step_gpu as T      # noqa: E402

dev = torch.device("cuda:0")
net, opt, step = T._make(dev, graph=False)
if len(sys.argv) > 1 and sys.argv[1] == "bf16":
    from npp_amd.model_augment import set_compute_dtype
    set_compute_dtype(torch.bfloat16)
size = int(os.environ.get("PROBE_SIZE", "64"))
im, lpar, lpose, w = T._batch(2, size, 3, dev)
for i in range(3):
    loss = float(step(im, lpar, lpose).detach())
    print("step", i, loss, flush=True)
print("PROBE_OK")
