"""Where do the D2D copies and fill launches of the supernet's weights pass come from?  One eager TrainStep-equivalent step under
torch.profiler; every copy / fill kernel is attributed to its innermost enclosing CPU op and autograd node."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NPP_STREAMS", "1")
os.environ["NPP_BENCH_SUPERVISE"] = "0"
import torch
from torch.profiler import profile, ProfilerActivity
import bench as B

NSTEPS = os.environ.get("NSTEPS", "1")
MODEL_ARGS = ["--model", "search", "--batch", "8"] if os.environ.get("MODEL", "search") == "search" else []
sys.argv = ["bench.py"] + MODEL_ARGS + ["--steps", NSTEPS, "--warmup", "1", "--graph", "0", "--no-cpu-baseline", "--no-prof"]
# run the bench's own set-up, profiling only its timed steps: bench.main() prints its line; we wrap the whole call
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False, with_stack=False) as prof:
    try:
        B.main()
    except SystemExit:
        pass
ev = prof.events()
by = collections.Counter()
WATCH = ("aten::copy_", "aten::zero_", "aten::fill_", "aten::clone", "aten::zeros", "aten::zeros_like", "aten::contiguous", "aten::_to_copy", "aten::cat",
         "aten::add", "aten::add_", "aten::mul", "aten::sum")
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name in WATCH:
        p = e.cpu_parent
        if p is not None and p.name in WATCH:
            continue      # (inner op of a watched op)
        chain = []
        while p is not None and len(chain) < 4:
            chain.append(p.name)
            p = p.cpu_parent
        by[(e.name, " < ".join(chain))] += 1
import json
json.dump({f"{n} | {ch}": c for (n, ch), c in by.items()}, open(f"gpurun_out/search_copies_{NSTEPS}.json", "w"))
