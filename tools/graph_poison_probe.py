"""Uninitialised / too-early reads that only a REPLAYED step can show (tests/test_poison_gpu.py::test_replayed_step_*).
Found with it (round 2): the second edge of a cell node that the hub topology runs on the hub stream read -- in its backward -- an
input whose block had gone back to its own stream's free list (no record_stream for the hub) and had become an f64 scratch of a later
BatchNorm backward: one 16x16 1x1 weight gradient of the 3x3-map decoder stage with elements of 1e35, invisible in the loss.  The memory of a hipGraph's private pool comes from fresh
hipMalloc segments, so the first replay reads whatever those pages held; later replays read the previous replay's values (finite,
nearly right -- invisible).  This probe makes the first case loud: after the eager warm-up step it fills most of the free HBM with
0xFF bytes (NaN as f32 / bf16), returns it to the driver and only then lets TrainStep capture, so the graph's pool is carved from
poisoned pages.  Modes as tests/comm_reducer_worker.py (SyncBatchNorm + GradReducer on a 1-rank RCCL group) or plain (local BN).
usage: graph_poison_probe.py [bf16] ; env PROBE_SYNC=0|1 PROBE_OVERLAP=0|1 PROBE_GB=<GiB to poison>"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29635")
import torch, torch.distributed as dist
from test_train_step_gpu import _batch, _cfg
from npp_amd import _ops as K, comm
from npp_amd.criterion import Criterion_par, Criterion_pose
from npp_amd.model_augment import Network, set_compute_dtype
from npp_amd.optim import FusedAdam
from npp_amd.train_step import TrainStep

search = os.environ.get("PROBE_MODEL", "augment") == "search"      # the supernet under SearchStep (two replayed passes)
sync = os.environ.get("PROBE_SYNC", "1") == "1" and not search
overlap = os.environ.get("PROBE_OVERLAP", "0") == "1"
dtype = torch.bfloat16 if "bf16" in sys.argv[1:] else torch.float32
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
set_compute_dtype(dtype)
torch.manual_seed(0)
red = None
if search:
    import test_train_step_gpu as T
    from npp_amd.train_step import SearchStep
    net, cp, cq, weights = T._search_setup(dev)
    if dtype == torch.bfloat16:
        set_compute_dtype(dtype)
    opt = FusedAdam(weights, lr=1e-3)
    a_opt = FusedAdam(net.arch_parameters(), lr=3e-3, betas=(0.5, 0.999), weight_decay=0.001)
    sstep = SearchStep(net, cp, cq, opt, a_opt, graph=True, warmup=1)
    b1, b2 = T._batch(2, 64, 3, dev), T._batch(2, 64, 4, dev)
else:
    net = Network(_cfg(8))
if sync:
    from npp_amd.ddp import GradReducer, unused_parameter_names
    dist.init_process_group("nccl", rank=0, world_size=1)
    comm.disable()
    K._SYNC_EVEN_ALONE = True
    net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
if search:
    def step(*_a):
        l1, l2 = sstep(b1[:3], b2[:3], entropy=True)
        return l1 + l2
    step.graph = None
    batch = ()
    opts = [opt, a_opt]
else:
    net = net.to(dev).train()
    cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
    opt = FusedAdam(list(net.parameters()) + list(cp.parameters()) + list(cq.parameters()), lr=1e-3)
    if sync:
        red = GradReducer(net, skip=unused_parameter_names(net), always_reduce=True, bucket_mb=0.5, overlap=overlap)
    step = TrainStep(net, cp, cq, opt, reducer=red, graph=os.environ.get("PROBE_GRAPH", "1") == "1", warmup=1)
    batch = _batch(2, 96, 5, dev)
    opts = [opt]
bad_total = 0
for it in range(4):
    if it == 1:      # the next call captures: poison the pages its pool will be carved from
        torch.cuda.synchronize()
        gb = float(os.environ.get("PROBE_GB", "24"))
        blocks = [torch.full((1 << 28,), -1, dtype=torch.int32, device=dev) for _ in range(int(gb))]      # 1 GiB of 0xFF each
        torch.cuda.synchronize()
        del blocks
        torch.cuda.empty_cache()
    loss = float(step(*batch))
    torch.cuda.synchronize()
    bg = [(n, tuple(p.shape)) for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    bp = [n for n, p in net.named_parameters() if not torch.isfinite(p).all()]
    bb = [n for n, b in net.named_buffers() if not torch.isfinite(b.float()).all()]
    names = {id(p): n for n, p in net.named_parameters()}
    # a gradient with a wild (but finite) element shows as exp_avg_sq = inf: g^2 overflows, the update is 0 and nothing else notices
    bs = [(names.get(id(p), "?"), k, int((~torch.isfinite(v)).sum()), v.numel()) for o in opts for p, st in o.state.items()
          for k, v in st.items() if torch.is_tensor(v) and v.is_floating_point() and not torch.isfinite(v).all()]
    wild = [(n, float(p.grad.abs().max())) for n, p in net.named_parameters() if p.grad is not None and float(p.grad.abs().max()) > 1e6]
    bad_total += len(bg) + len(bp) + len(bb) + int(loss != loss) + len(bs) + len(wild)
    if bs or wild:
        print("   adam state non-finite:", bs[:6], "wild gradients:", wild[:6], flush=True)
    print("step", it, "loss", loss, "graphed", (sstep.weights_pass.graph is not None) if search else (step.graph is not None), "non-finite grads", len(bg), bg[:6], "params", len(bp), bp[:4],
          "buffers", len(bb), bb[:4], flush=True)
try:      # (SyncBatchNorm over the mailboxes: how many exchanges ran inside the fused kernels, and whether a mailbox reported an error)
    from npp_amd import _ops as _K
    from npp_amd import comm as _comm
    print("folded exchanges (forward, backward launches):", _K.FOLD_STATS, "p2p", "ok" if _comm.p2p_ok() else "ERROR", "active", _comm.p2p_active(), flush=True)
    if _comm.p2p_active() and not _comm.p2p_ok():
        bad_total += 1
except Exception as e:      # noqa: BLE001
    print("fold stats unavailable:", e)
print("GRAPH_PROBE_DONE" if bad_total == 0 else "GRAPH_PROBE_BAD")
if sync:
    dist.destroy_process_group()
