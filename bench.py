"""bench.py -- img/s of the NPPNet (model_augment, fixed genotype, C=64, L=16, R=1) training step on MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch 16] [--size 384] [--dtype bf16|f32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One step = model(images) -> Criterion_par + Criterion_pose -> mean -> zero_grad -> backward (+ bucketed RCCL
gradient all-reduce, SyncBN statistics exchange when N > 1) -> Adam.step  (core/function.py:87-107), on a
synthetic LIP-shaped batch that is already resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FWD_GFLOP_PER_IMG = {384: 243.37, 512: 432.65}   # BASELINE.md §2 (conv FLOPs, 2*MAC)
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}      # MI355X_MICROARCH.md chip table (dense)


def cfg_ns(C=64):
    from types import SimpleNamespace as NS
    return NS(DATASET=NS(NUM_CLASSES=20, NUM_JOINTS=16), TRAIN=NS(LAYERS=16, INIT_CHANNELS=C),
              MODEL=NS(DECONV_WITH_BIAS=False, HEAD='PSP', REFINE_LAYERS=1))


def cpu_baseline(size, budget_s=25.0):
    """The CPU oracle (a port of the reference path, oracle/nppnet_oracle.py) timed on this host's cores:
    N=1 train-mode fwd + both losses + bwd, fp32.  Baseline only -- not the thing shipped or measured."""
    import torch
    from oracle import nppnet_oracle as O
    from npp_amd.model_augment import Network
    from npp_amd.synth import synth_batch
    torch.manual_seed(0)
    net = Network(cfg_ns())
    tensors = {k: v.detach().clone() for k, v in net.state_dict().items()}
    for k, v in tensors.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    del net
    images, lpar, lpose, _ = synth_batch(1, size, seed=0)
    images = torch.from_numpy(images)
    lpar = [torch.from_numpy(a) for a in lpar]
    lpose = [torch.from_numpy(a[:, :-1].copy()) for a in lpose]
    lam_pose = torch.full((2,), -2.5, requires_grad=True)
    lam_par = torch.full((2,), 2.3, requires_grad=True)
    def one():
        for v in tensors.values():
            v.grad = None
        loss, _, _, _ = O.train_step_loss(tensors, images, lpar, lpose, lam_pose, lam_par)
        loss.backward()

    # pick the thread count that is fastest on this host (oneDNN over-subscribes badly on 100+ core boxes)
    ncpu = os.cpu_count() or 8
    best_t, best_dt = None, None
    one()
    sweep = {}
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64, ncpu)}):      # (ncpu = BASELINE.md section 4's setting)
        if best_dt is not None and nt > 64 and sweep[max(sweep)] > 1.5 * best_dt:
            sweep[nt] = "not tried (already 1.5x slower than the best at %d threads)" % max(k for k in sweep if isinstance(k, int))
            continue
        torch.set_num_threads(nt)
        t0 = time.time()
        one()
        dt = time.time() - t0
        _hb(f"cpu baseline: {nt} threads tried")      # (the supervisor of the 1-GPU run watches for progress)
        sweep[nt] = round(dt, 2)
        if best_dt is None or dt < best_dt:
            best_t, best_dt = nt, dt
    torch.set_num_threads(best_t)
    times = []
    t_start = time.time()
    it = 1
    while True:
        t0 = time.time()
        for v in tensors.values():
            v.grad = None
        loss, _, _, _ = O.train_step_loss(tensors, images, lpar, lpose, lam_pose, lam_par)
        loss.backward()
        dt = time.time() - t0
        if it >= 1:
            times.append(dt)
        _hb("cpu baseline: iteration done")
        it += 1
        if (time.time() - t_start > budget_s and len(times) >= 2) or len(times) >= 6:
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(1.0 / med, 4), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"N=1 {size}x{size} fp32 train-mode fwd+losses+bwd, median of {len(times)} iterations after "
                      f"{1 + len(sweep)} warm-up iterations (torch-CPU oracle, {os.cpu_count()} logical CPUs; `cores` = the "
                      f"fastest of the thread counts tried, seconds per iteration by threads: {sweep})"}


def spawn_ranks(n):
    """Run this script as `n` fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), relay
    rank 0's stdout (its last line is the JSON record) and return the worst exit code.  Called before torch is imported:
    the parent never initialises the GPU, and nothing is re-executed in a process that has."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(s.getsockname()[1])
        s.close()
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    import tempfile
    procs = []
    with tempfile.TemporaryFile() as out0:
        for r in range(n):
            e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        # a rank that dies leaves the others waiting in a collective: end them (by their own PIDs) instead of hanging
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):
                time.sleep(5.0)
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                break
            time.sleep(0.2)
        rcs = [p.wait() for p in procs]
        out0.seek(0)
        sys.stdout.write(out0.read().decode(errors="replace"))
        sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write(f"bench.py: ranks failed (rank, exit code): {bad}\n")
        return max(abs(rc) for _, rc in bad) or 1
    return 0


def _hb(msg):
    """Progress mark for the rank supervisor (supervise_rank): one line per phase in the heartbeat file."""
    path = os.environ.get("NPP_BENCH_HB")
    if path:
        try:
            with open(path, "a") as fh:
                fh.write(f"{time.time():.1f} {msg}\n")
        except OSError:
            pass


def _tiers_multi():
    """Attempts of an N > 1 rank, see supervise_rank."""
    return [
        ({}, [], None),
        ({"NPP_P2P_FOLD": "0"}, [],
         "the first attempt (peer-to-peer SyncBatchNorm exchange inside the fused BatchNorm kernels) died or stalled; this line is from "
         "fresh workers with every exchange a stand-alone peer-to-peer kernel (the round-4 form)"),
        ({"NPP_SYNCBN_P2P": "0"}, [],
         "the peer-to-peer attempts died or stalled; this line is from fresh workers with every exchange "
         "as a collective of the process group"),
        ({"NPP_SYNCBN_P2P": "0"}, ["--graph", "0"], "the hipGraph attempts died or stalled; this line is from a fresh eager (--graph 0) worker"),
    ]


def _tiers_single(queues="2"):
    """Attempts of the 1-GPU run: the replayed step on TWO hardware queues (GPU_MAX_HW_QUEUES=2: the hipGraph executor then keeps the
    two branch chains on one queue each instead of spreading their segments over three -- 45.4 vs 45.9 ms per step, measured A/B/A on
    one box; 8 queues: 74.5 ms), then the runtime's default in a fresh process should the first one not deliver (a capture on fewer
    queues than captured streams has been seen to crash the runtime: the supernet's step at 2 queues, which therefore gets 3 --
    68.8 ms per step against 71.1 at the default of 4 and 238 at 6)."""
    return [
        ({"GPU_MAX_HW_QUEUES": queues}, [], None),
        ({}, [], f"the first worker (GPU_MAX_HW_QUEUES={queues}) died or stalled; this line is from a fresh worker on the runtime's default queues"),
    ]


def supervise_rank(tiers=None):
    """One rank of an N > 1 run (under torchrun, or a child of spawn_ranks): this process never touches the GPU.  It starts the
    real worker as a CHILD (NPP_BENCH_WORKER=1), watches its heartbeat file, and if the worker dies or makes no progress for
    NPP_BENCH_STALL_S seconds (a hipGraph capture of RCCL collectives that hangs at N > 1 has never been seen OR ruled out: no
    multi-GPU box was available to the builder) kills exactly that child and starts a FRESH one on a fresh rendezvous port -- first with the
    SyncBatchNorm exchanges as collectives instead of the peer-to-peer kernel, then eager (--graph 0) -- so that the first real 8-GPU run cannot end with no line at all.  Rank 0's worker writes its JSON record to
    a result file as soon as the timed region is over; a worker that hangs after that (in the extra communicator queries) does
    not cost the measurement."""
    import subprocess
    import tempfile
    rank = int(os.environ.get("RANK", "0"))
    base_port = int(os.environ.get("MASTER_PORT", "29511"))
    stall_first = float(os.environ.get("NPP_BENCH_STALL_FIRST_S", "420"))      # first progress mark: torch import on a cold box
    stall = float(os.environ.get("NPP_BENCH_STALL_S", "240"))
    tmp = tempfile.mkdtemp(prefix=f"npp_bench_r{rank}_")
    last_rc = 1
    # attempt 0: the default step (hipGraph, SyncBatchNorm statistics through the peer-to-peer mailboxes of csrc/p2p.hip, exchanged inside
    #            the fused BatchNorm kernels); attempt 1: the same with stand-alone exchange kernels (NPP_P2P_FOLD=0);
    # attempt 2: the same with every exchange as an RCCL collective on the hub stream (the round-2 form: neither transport has run on
    #            a multi-GPU box the builder had access to, so neither may be the only one); attempt 3: collectives, eager (--graph 0)
    if tiers is None:
        tiers = _tiers_multi()
    for attempt, (tier_env, tier_argv, tier_note) in enumerate(tiers):
        hb = os.path.join(tmp, f"hb{attempt}")
        res = os.path.join(tmp, f"result{attempt}.json")
        open(hb, "w").close()
        env = dict(os.environ, NPP_BENCH_WORKER="1", NPP_BENCH_HB=hb, NPP_BENCH_RESULT=res, NPP_BENCH_ATTEMPT=str(attempt),
                   MASTER_PORT=str(base_port + 211 + attempt))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)      # the workers' rank 0 hosts the store of ITS attempt's port
        argv = list(sys.argv[1:])
        env.setdefault("NPP_P2P_TIMEOUT_MS", "8000")        # a peer that never shows up is reported after 8 s, not 20
        env.update(tier_env)
        argv += tier_argv
        if tier_note:
            env["NPP_BENCH_FALLBACK"] = tier_note
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                 stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL)
        t_last, size_last, seen_any = time.time(), 0, False
        stalled = False
        while child.poll() is None:
            time.sleep(0.5)
            try:
                sz = os.path.getsize(hb)
            except OSError:
                sz = size_last
            if sz != size_last:
                size_last, t_last, seen_any = sz, time.time(), True
            if time.time() - t_last > (stall if seen_any else stall_first):
                stalled = True
                sys.stderr.write(f"bench.py supervisor (rank {rank}): worker {child.pid} made no progress for "
                                 f"{time.time() - t_last:.0f} s (attempt {attempt}); ending it\n")
                child.kill()
                break
        out = child.stdout.read().decode(errors="replace") if child.stdout is not None else ""
        last_rc = child.wait()
        have_result = os.path.isfile(res) and os.path.getsize(res) > 0
        if last_rc == 0 and not stalled:
            if rank == 0:
                sys.stdout.write(out)
                sys.stdout.flush()
            return 0
        if have_result:      # timed region done, the worker was lost afterwards: its record stands
            if rank == 0:
                with open(res) as fh:
                    sys.stdout.write(fh.read().strip() + "\n")
                sys.stdout.flush()
            return 0
        sys.stderr.write(f"bench.py supervisor (rank {rank}): attempt {attempt} ended with rc {last_rc}"
                         f"{' (stalled)' if stalled else ''}\n")
    return abs(last_rc) or 1


def _under_profiler() -> bool:
    """rocprofv3 (or any rocprofiler-sdk tool) is attached to THIS process: its preloaded library has already initialised the GPU
    here, so the process must launch the kernels itself -- starting the worker as a child would be the fork+exec from a
    GPU-initialised process that this pool forbids, and the profiled process would no longer be the one that runs them."""
    env = os.environ
    if any(k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER", "ROCTX")) for k in env):
        return True
    pre = env.get("LD_PRELOAD", "") + ":" + env.get("HSA_TOOLS_LIB", "")
    return any(s in pre for s in ("rocprofiler", "rocprof", "roctracer", "librocp"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("NPP_BENCH_BATCH", "16")), help="images per GPU")
    ap.add_argument("--size", type=int, default=384)
    ap.add_argument("--dtype", default=os.environ.get("NPP_BENCH_DTYPE", "bf16"), choices=["bf16", "f32"])
    ap.add_argument("--model", default="augment", choices=["augment", "search"],
                    help="augment: model_augment.Network (the metric's workload); search: the MixedOp supernet of "
                         "BASELINE config 5 (C=32, weights-only train() pass)")
    ap.add_argument("--alpha-pass", action="store_true",
                    help="--model search: time the whole train_with_alpha iteration (core/function.py:485-621): the weights pass "
                         "AND the architecture pass on a second batch (npp_amd.train_step.SearchStep)")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam(fused=True) instead of npp_amd.optim.FusedAdam")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--local-bn", action="store_true", help="ablation: do not synchronise BN statistics across ranks")
    ap.add_argument("--force-dist", action="store_true",
                    help="exercise the multi-GPU path (process group, SyncBN exchange, bucketed all-reduce) on ONE rank")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("NPP_BENCH_GRAPH", "-1")),
                    help="1: capture the whole step (fwd+loss+bwd+Adam) in one hipGraph and replay it; 0: eager; "
                         "-1 (default): graph for every N (on N > 1 every collective sits on the capture's origin stream); a capture "
                         "failure falls back to eager")
    ap.add_argument("--launcher", default="", choices=["", "eager", "auto"],
                    help="time the loop body exactly as the UNCHANGED reference launcher issues it (core/function.py:87-107: "
                         "model(images), criteria, zero_grad, backward, torch.optim.Adam.step) instead of TrainStep: "
                         "eager = kernel by kernel, auto = with npp_amd.auto_graph (NPP_AUTO_GRAPH=1)")
    ap.add_argument("--no-comm-ablation", action="store_true",
                    help="N > 1: skip the second, communication-free timing (local BN statistics, no gradient reducer) "
                         "that `exposed_comm_ms` is computed from")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher: start the N ranks ourselves (one process per GPU, as the reference's
        # launcher does: augment_lip_sync.py:107-113, README.md:14).  This parent has made no GPU call and makes none.
        raise SystemExit(spawn_ranks(args.gpus))
    if _under_profiler():      # (the supervisor must never spawn from a process the profiler has GPU-initialised)
        os.environ["NPP_BENCH_SUPERVISE"] = "0"
    if (args.gpus > 1 and not os.environ.get("NPP_BENCH_WORKER") and os.environ.get("NPP_BENCH_SUPERVISE", "1") != "0"):
        raise SystemExit(supervise_rank())      # a rank of an N > 1 run: watchdog parent + worker child (see supervise_rank)
    if (args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1") or "1") == 1 and not os.environ.get("NPP_BENCH_WORKER")
            and os.environ.get("NPP_BENCH_SUPERVISE", "1") != "0" and "GPU_MAX_HW_QUEUES" not in os.environ
            and not args.force_dist and not args.launcher and not args.alpha_pass):
        # the default 1-GPU line: the worker runs as a child on two hardware queues, with a fresh default-queue worker behind it
        raise SystemExit(supervise_rank(_tiers_single("2" if args.model == "augment" else "3")))
    _hb("worker started")
    fake = os.environ.get("NPP_BENCH_FAKE_WORKER")
    if fake:
        # tests/test_bench_supervisor_cpu.py: the watchdog's logic without a GPU.  "hang": the first attempt stops making progress
        # (as a hung capture would), the --graph 0 attempt answers; "late_hang": the record is written, then the worker hangs
        attempt = int(os.environ.get("NPP_BENCH_ATTEMPT", "0"))
        rec = {"metric": "fake", "value": 1.0, "graph_arg": args.graph, "attempt": attempt}
        if os.environ.get("NPP_BENCH_FALLBACK"):
            rec["fallback"] = os.environ["NPP_BENCH_FALLBACK"]
        rec["syncbn_p2p"] = os.environ.get("NPP_SYNCBN_P2P", "1")
        rec["p2p_fold"] = os.environ.get("NPP_P2P_FOLD", "1")
        rec["hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES")
        if fake == "hang" and attempt == 0:
            time.sleep(3600)
        if fake == "hang2" and attempt <= 1:
            time.sleep(3600)
        if fake == "hang3" and attempt <= 2:
            time.sleep(3600)
        if fake == "late_hang":
            with open(os.environ["NPP_BENCH_RESULT"], "w") as fh:
                fh.write(json.dumps(rec) + "\n")
            time.sleep(3600)
        if fake == "crash" and attempt == 0:
            raise SystemExit(3)
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps(rec), flush=True)
        return

    import torch
    import torch.distributed as dist
    _hb("torch imported")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("NPP_BENCH_ONE_GPU"):      # rehearsal of the N > 1 flow on a 1-GPU box: every rank on device 0 (with
        local_rank = 0                           # NPP_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device)
        # the ranks share ONE chip: the workgroups that wait inside a kernel for another rank's statistics (csrc/p2p_xp.h) must leave
        # room for that rank's kernels -- the budget of one rank (192) over the world's ranks
        os.environ.setdefault("NPP_XP_MAX_BLOCKS", str(max(32, 192 // max(1, int(os.environ.get("WORLD_SIZE", "1"))))))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("NPP_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    _hb("process group up")
    from npp_amd import _lib
    from npp_amd.model_augment import Network, set_compute_dtype
    from npp_amd.criterion import Criterion_par, Criterion_pose
    from npp_amd.synth import synth_batch
    from npp_amd.ddp import GradReducer, unused_parameter_names

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    set_compute_dtype(dtype)
    torch.manual_seed(0)
    if args.model == "search":
        from types import SimpleNamespace as NS
        from npp_amd.model_search_interact import Network as SearchNetwork
        net = SearchNetwork(NS(DATASET=NS(NUM_CLASSES=20, NUM_JOINTS=16), SEARCH=NS(LAYERS=16, INIT_CHANNELS=32),
                               MODEL=NS(DECONV_WITH_BIAS=False, HEAD='PSP', REFINE_LAYERS=1)))
        args.no_cpu_baseline = True      # the CPU baseline leg times the augment network only
    else:
        net = Network(cfg_ns())
    sync_bn = use_dist and not args.local_bn
    if args.force_dist:
        from npp_amd import _ops as _K
        _K._SYNC_EVEN_ALONE = True
    if sync_bn:
        net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)   # augment_lip_sync.py:191
    net = net.to(dev).train()
    crit_pose = Criterion_pose(out_len=2).to(dev)
    crit_par = Criterion_par(out_len=2).to(dev)
    use_graph = args.graph != 0      # default: graph for every N (RCCL collectives are captured too; falls back to eager)
    # buckets go out as they complete: eagerly on the reducer's side stream, under capture on the capture's origin stream
    # (ddp.GradReducer._target_stream, _ops.hub_all_reduce)
    # Buckets are all-reduced by finish(), after backward (overlap=False): the step can then defer and batch its weight gradients
    # straight into the bucket slots (1-rank exercise: 57.8 vs 62.4 ms with the buckets issued from the hooks during backward; on 8
    # GPUs the 308 MB all-reduce is then exposed, ~2 ms on xGMI).  NPP_DDP_OVERLAP=1 restores the overlapped form.
    # N > 1: BOTH forms are timed for a few steps below (`overlap_ab`) and the faster one runs the timed region; the overlapped form
    # (north_star) is the default wherever no such measurement exists (1 rank: --force-dist) -- NPP_DDP_OVERLAP=0 / 1 forces either.
    # Round 3: the overlapped form is "tail" -- the step keeps its batched weight-gradient tail and the buckets of the KxK conv weights
    # (4/5 of the bytes) are reduced under the tail's second group (ddp.GradReducer, train_step.TrainStep._tail_with_reducer); the
    # round-2 form (buckets issued from the gradient hooks during backward, every weight gradient a launch of its own: +9 ms on one
    # rank) is NPP_DDP_OVERLAP=hooks.
    overlap_env = os.environ.get("NPP_DDP_OVERLAP")
    overlap_pick = {"0": False, "1": True, "hooks": True, "tail": "tail"}.get(overlap_env, "tail") if overlap_env is not None else "tail"
    overlap_ab = None

    def make_reducer(overlap):
        return GradReducer(net, skip=unused_parameter_names(net), always_reduce=args.force_dist, overlap=overlap)

    reducer = make_reducer(overlap_pick) if use_dist else None
    arch_ids = {id(a) for a in net.arch_parameters()} if args.model == "search" else set()
    params = [q for q in net.parameters() if id(q) not in arch_ids] + list(crit_pose.parameters()) + list(crit_par.parameters())
    # Adam (augment_lip_sync.py:210-213).  Default: npp_amd.optim.FusedAdam, one launch over a device job table (SURVEY
    # §8f-2); --torch-adam: torch's fused multi-tensor Adam (~46 launches, 1.8 ms/step); the capturable foreach path
    # issues ~3000 scalar-math launches per step (13 ms)
    if args.torch_adam:
        try:
            opt = torch.optim.Adam(params, lr=1e-4, fused=True, capturable=use_graph)
        except (RuntimeError, ValueError):
            opt = torch.optim.Adam(params, lr=1e-4, capturable=use_graph)
    else:
        from npp_amd.optim import FusedAdam
        opt = FusedAdam(params, lr=1e-4)

    images, lpar, lpose, _ = synth_batch(args.batch, args.size, seed=0, rank=rank)
    images = torch.from_numpy(images).to(dev)
    lpar = [torch.from_numpy(a).to(dev) for a in lpar]
    lpose = [torch.from_numpy(a[:, :-1].copy()).to(dev) for a in lpose]

    # the step is the product's own callable (npp_amd/train_step.py; core/function.py:72-107): eager calls first, then one
    # hipGraph capture of the static step (~5000 launches and, for N > 1, the SyncBN and gradient collectives
    # become one replay); any capture failure falls back to eager on every rank
    from npp_amd.train_step import TrainStep
    from npp_amd import _ops as K
    if args.model == "search" and args.alpha_pass:
        from npp_amd.train_step import SearchStep
        from npp_amd.optim import FusedAdam as _FA
        if reducer is not None:
            raise SystemExit("--alpha-pass is a single-GPU bench line (the N > 1 search run times the weights pass)")
        a_opt = _FA(net.arch_parameters(), lr=3e-4, betas=(0.5, 0.999), weight_decay=0.001)     # search_lip_sync.py:279
        im2, lpar2, lpose2, _ = synth_batch(args.batch, args.size, seed=1, rank=rank)
        batch2 = (torch.from_numpy(im2).to(dev), [torch.from_numpy(a).to(dev) for a in lpar2],
                  [torch.from_numpy(a[:, :-1].copy()).to(dev) for a in lpose2])
        search_step = SearchStep(net, crit_pose, crit_par, opt, a_opt, graph=use_graph, warmup=2)
        train_step = search_step.weights_pass

        def step():
            return search_step((images, lpar, lpose), batch2)[0]
    else:
        if use_dist and world > 1 and overlap_env is None and not args.launcher:
            # A/B of the two reducer forms on THIS node (a 1-rank run cannot decide it: there is no wire)
            ab_steps = max(2, min(5, args.steps))
            overlap_ab = {}
            for mode in ("tail", False):
                if reducer is not None:
                    reducer.remove()
                reducer = make_reducer(mode)
                ts = TrainStep(net, crit_pose, crit_par, opt, reducer=reducer, graph=use_graph, warmup=2)
                for _ in range(3 if use_graph else 1):
                    ts(images, lpar, lpose)
                dist.barrier()
                torch.cuda.synchronize()
                t_ab = time.perf_counter()
                for _ in range(ab_steps):
                    ts(images, lpar, lpose)
                dist.barrier()
                torch.cuda.synchronize()
                t_ab = torch.tensor([time.perf_counter() - t_ab], dtype=torch.float64, device=dev)
                dist.all_reduce(t_ab, op=dist.ReduceOp.MAX)
                overlap_ab["overlapped_ms" if mode else "after_backward_ms"] = round(float(t_ab) / ab_steps * 1e3, 3)
                overlap_ab["overlapped_graph" if mode else "after_backward_graph"] = bool(ts.graphed)
                _hb(f"overlap A/B: mode {mode} timed")
                del ts
            overlap_pick = "tail" if overlap_ab["overlapped_ms"] <= overlap_ab["after_backward_ms"] else False
            overlap_ab["picked"] = "overlapped (under the weight-gradient tail)" if overlap_pick else "after_backward"
            overlap_ab["steps_each"] = ab_steps
            reducer.remove()
            reducer = make_reducer(overlap_pick)
        train_step = TrainStep(net, crit_pose, crit_par, opt, reducer=reducer, graph=use_graph, warmup=2)

        def step():
            return train_step(images, lpar, lpose)

    def eager_step():
        return train_step._eager(images, lpar, lpose, None)

    if args.launcher:
        from npp_amd import auto_graph
        auto_graph.ENABLED = args.launcher == "auto"
        net._auto_graph_off = False
        use_graph = False
        args.no_prof = True
        K.GRAPH_TOPOLOGY = False                        # (AutoGraph switches it on itself)
        l_opt = torch.optim.Adam(params, lr=1e-4)       # augment_lip_sync.py:210-213
        if os.environ.get("NPP_LAUNCHER_FUSED_ADAM"):      # INTEGRATION.md's optional one-line change
            from npp_amd.optim import FusedAdam as _FA2
            l_opt = _FA2(params, lr=1e-4)

        host = {"forward": 0.0, "criteria": 0.0, "zero_grad": 0.0, "backward": 0.0, "optimizer": 0.0, "n": 0}

        def step():      # (host-side enqueue time per section is kept for NPP_LAUNCHER_HOST_TIMES=1)
            t0 = time.perf_counter()
            output_pose, output_par = net(images)
            t1 = time.perf_counter()
            loss = (crit_par(output_par, lpar).unsqueeze(0) + crit_pose(output_pose, lpose).unsqueeze(0)).mean()
            t2 = time.perf_counter()
            l_opt.zero_grad()
            t3 = time.perf_counter()
            loss.backward()
            if reducer is not None:
                reducer.finish()
            t4 = time.perf_counter()
            l_opt.step()
            t5 = time.perf_counter()
            for k, a, b in (("forward", t0, t1), ("criteria", t1, t2), ("zero_grad", t2, t3), ("backward", t3, t4), ("optimizer", t4, t5)):
                host[k] += b - a
            host["n"] += 1
            return loss

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    _hb("model built")
    if use_graph or args.launcher == "auto":
        for _ in range(3):      # two eager calls, then capture + first replay: all before the W warm-up steps
            step()
            _hb("pre-warm-up step issued")
        barrier()
        _hb("captured")
    graph = train_step.graph

    for _ in range(args.warmup):
        step()
    barrier()
    if use_dist:
        # the peer-to-peer exchange reports a peer that never answered through an error word, not by hanging: every rank must have
        # seen every peer in the warm-up steps, else this attempt ends here and the supervisor starts the collective form
        from npp_amd import comm as _cm
        okf = torch.tensor([1 if _cm.p2p_ok() else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if int(okf.item()) != 1:
            sys.stderr.write(f"bench.py (rank {rank}): a peer-to-peer SyncBatchNorm exchange timed out during warm-up; giving up this attempt\n")
            raise SystemExit(17)
        _hb("peers all present")
    prof = not args.no_prof
    L = _lib.lib()
    import ctypes as C
    dt_code = _lib.NPP_BF16 if dtype == torch.bfloat16 else _lib.NPP_F32
    if args.launcher:
        for k in host:
            host[k] = 0.0 if k != "n" else 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    _hb("timed region done")
    ranks_identical = None
    if use_dist:
        # Rank-order sums of the SyncBatchNorm statistics and averaged gradient buckets must leave every rank with bit-identical
        # parameters and BatchNorm buffers: a stale or torn statistic of the peer-to-peer exchange, or a rank that used local
        # statistics, shows up HERE and nowhere else (the loss would look plausible).  Two position-weighted sums of the raw bit
        # patterns per group, MIN == MAX over the ranks.
        def bit_sums(tensors):
            a = torch.zeros(2, dtype=torch.int64, device=dev)
            for t in tensors:
                t = t.detach().reshape(-1)
                if t.numel() == 0:
                    continue
                bits = t.view(torch.int32).to(torch.int64) if t.element_size() == 4 else (t.to(torch.int64) if not t.is_floating_point() else t.double().view(torch.int64))
                w = (torch.arange(bits.numel(), device=dev, dtype=torch.int64) % 1021) + 1
                a[0] += bits.sum()
                a[1] += (bits * w).sum()
            return a
        mdl = net.module if hasattr(net, "module") else net
        sums = torch.cat([bit_sums(list(mdl.parameters())), bit_sums(list(mdl.buffers()))])
        lo, hi = sums.clone(), sums.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        same = (lo == hi).tolist()
        ranks_identical = {"parameters": bool(same[0] and same[1]), "bn_buffers": bool(same[2] and same[3]),
                           "how": "MIN == MAX over the ranks of two position-weighted sums of the raw bit patterns, after the timed steps"}
        _hb("rank identity checked")
    if args.launcher and os.environ.get("NPP_LAUNCHER_HOST_TIMES") and rank == 0:
        sys.stderr.write("host enqueue ms per step (timed region): " + ", ".join(
            f"{k} {1e3 * v / max(host['n'], 1):.2f}" for k, v in host.items() if k != "n") + "\n")
    # roofline of the dominant kernel: HIP events around every launch of the two MFMA conv families (on the launch stream), in
    # eager single-stream steps right after the timed region -- a replayed graph cannot carry the event pairs, and with the
    # two branch streams overlapping an event pair would also time the neighbour's kernels.  The family with the larger
    # total time is reported as `roofline`, the other one next to it.
    roof = None
    roofs = []
    if prof:
        streams_env = os.environ.get("NPP_STREAMS")
        os.environ["NPP_STREAMS"] = "1"
        prof_steps = 2
        KERNELS = {"conv_s1": "conv_s1_kernel (stride-1 KxK conv fwd + dgrad, LDS-resident footprint, MFMA 32x32x16)",
                   "conv_g8": "conv_g8_kernel (1x1 conv fwd + dgrad: 8-phase LDS-DMA implicit GEMM, MFMA 16x16x32)",
                   "conv_g4": "conv_g4_kernel + conv_h3_kernel + conv_c32_kernel + conv_thin (conv fwd + dgrad, incl. the merged edges of round 4: same-input edges as one conv C -> mC and one data gradient mC -> C; g4: 1x1 / 3x3 on 64x64 / 128x128 / 64x32 tiles, stride 1 and 2, LDS-DMA ring of 2; h3: 3x3 with an LDS-resident halo footprint on the >= 30k-pixel maps; c32: 32 -> 32 3x3 with halo and all nine taps resident, output / input groups; MFMA 16x16x32; 114 of the data-gradient launches per step also deliver the BatchNorm-backward sums of the gradient they finish -- two more tensor reads each, timed here, not in the algorithmic flops: the family read 0.156 before that fusion)",
                   "conv_wgrad": "weight-gradient family (every LDS-DMA-eligible weight gradient of the step in one launch per kernel variant over a device job table -- conv_wgrad_g9_batched_kernel (round 5): 3x3 layers on maps of whole 8x16-pixel tiles, all nine taps from one LDS-resident halo, 8 waves, slabs summed by the batched unpack; conv_wgrad_g4_batched_kernel: 128x128 tile of one tap, the 1x1 and small-map layers; conv_wgrad_narrow_batched_kernel: 32 / 64-channel 3x3; all pixel-major LDS-DMA (inline asm) + transposing reads, MFMA 16x16x32; strided / odd-channel layers on conv_wgrad_kernel)"}
        # HBM bytes per launch: rocprofv3 --pmc cannot run inside this process, so the counters come from the committed passes
        # of this same command (tools/final_profiles.sh -> profiles/rNN_pmc_traffic.json, the newest round's) -- and only while the kernel sources
        # still hash to what those passes ran on; otherwise `traffic` is null and `traffic_note` says why
        traffic_db, traffic_note = {}, None
        import glob
        cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")))      # the newest round's counters
        traffic_file = os.path.relpath(cands[-1], REPO) if cands else os.path.join("profiles", "r04_pmc_traffic.json")
        try:
            with open(os.path.join(REPO, traffic_file)) as fh:
                traffic_db = json.load(fh)
            if traffic_db.get("source_hash") != _lib.kernel_source_hash():
                traffic_note = (f"{traffic_file} was collected on kernel sources {traffic_db.get('source_hash')}, this build is "
                                f"{_lib.kernel_source_hash()}: stale, not reported (re-run tools/final_profiles.sh)")
                traffic_db = {}
            elif _lib.built_source_hash() != _lib.kernel_source_hash():
                traffic_note = (f"libnpp_hip.so was built from sources {_lib.built_source_hash()}, the tree holds "
                                f"{_lib.kernel_source_hash()}: the counters in {traffic_file} are not this library's, not reported")
                traffic_db = {}
        except Exception as e:      # noqa: BLE001
            traffic_db, traffic_note = {}, f"{traffic_file}: {e}"
        # the same families' kernel time inside the REPLAYED graph (two branch streams overlapping, rocprofv3 --kernel-trace --stats of this
        # command): from the committed summary of the newest round, under the same source-hash rule as the traffic
        graph_db = {}
        try:
            gcands = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_family_graph_ms.json")))
            if gcands:
                with open(gcands[-1]) as fh:
                    gd = json.load(fh)
                if gd.get("source_hash") == _lib.kernel_source_hash() == _lib.built_source_hash():
                    graph_db = gd
                    graph_db["_file"] = os.path.relpath(gcands[-1], REPO)
        except Exception:      # noqa: BLE001
            graph_db = {}
        for fam, label in KERNELS.items():
            _hb(f"roofline leg: {fam}")
            # one eager step per measurement, the faster of `prof_steps` measurements is reported: a single stalled launch (seen once:
            # conv_g8 at 3x its usual time in one of two steps, rocprofv3 of the same run showing nothing) must not set the figure
            best = None
            for _ in range(prof_steps):
                L.npp_prof_begin(_lib.FAM[fam], dt_code)
                eager_step()
                barrier()
                ms, fl, by, nl = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
                L.npp_prof_end(C.byref(ms), C.byref(fl), C.byref(by), C.byref(nl))
                if nl.value > 0 and ms.value > 0 and (best is None or ms.value < best[0].value):
                    best = (ms, fl, by, nl)
            if best is not None:
                ms, fl, by, nl = best
                ach = fl.value / (ms.value * 1e-3) / 1e12
                peak = PEAK_TFLOPS[args.dtype]
                rec = traffic_db.get(fam)      # the family's own average: every launch of every kernel the family times
                traffic = round(rec["traffic_bytes_per_launch"]) if rec and rec.get("launches") else None
                roofs.append({"bound": "mfma", "kernel": label, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                              "frac": round(ach / peak, 4), "traffic": traffic,
                              "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 PMC)",
                              "traffic_source": traffic_note or f"tools/final_profiles.sh -> {traffic_file} (sources {traffic_db.get('source_hash')})",
                              "launches_per_step": nl.value, "ms_per_step": round(ms.value, 3),
                              "avg_launch_us": round(ms.value * 1e3 / nl.value, 2),
                              "algorithmic_gflop_per_launch": round(fl.value / nl.value / 1e9, 4),
                              "algorithmic_mbyte_per_launch": round(by.value / nl.value / 1e6, 3)})
                grec = graph_db.get(fam)
                if grec and grec.get("ms_per_step"):      # (the event-timed figure above is the eager single-stream one)
                    roofs[-1]["graph_replay"] = {"ms_per_step": grec["ms_per_step"], "launches_per_step": grec.get("launches_per_step"),
                                                 "frac": round(fl.value / (grec["ms_per_step"] * 1e-3) / 1e12 / peak, 4),
                                                 "source": f"rocprofv3 --kernel-trace --stats of the replayed hipGraph run, {graph_db.get('_file')}"}
        roofs.sort(key=lambda r: -r["ms_per_step"])
        roof = roofs[0] if roofs else None
        if streams_env is None:
            os.environ.pop("NPP_STREAMS", None)
        else:
            os.environ["NPP_STREAMS"] = streams_env
    comm = None
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        # what the transport itself reports: one all-reduce of ones through the process group the step used (= the number of
        # ranks that took part), and -- further down, after the record is safe -- ncclCommCount of an RCCL communicator
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(ones)
        comm = {"backend": dist.get_backend(), "rccl_ranks": int(round(float(ones))), "allreduce_of_ones": float(ones),
                "rccl_ranks_source": "all-reduce of ones through the step's process group",
                "ddp_overlap": (("tail" if getattr(reducer, "tail", False) else bool(reducer.overlap)) if reducer is not None else None)}
        if overlap_ab is not None:
            comm["overlap_ab"] = overlap_ab
        from npp_amd import comm as _cm
        comm["syncbn_transport"] = ("p2p mailboxes (csrc/p2p.hip, %s memory, %s 8-byte {data|tag} units: the mode that passed the %d-exchange acceptance test), %d exchanges issued by the host, peers %s" %
                                    (("uncached", "fine-grained", "plain device")[max(0, min(2, int(_lib.lib().npp_p2p_alloc_kind())))],
                                     _cm._p2p.get("mode"), _cm.SELFTEST_EXCHANGES, _cm._p2p["count"], "all present" if _cm.p2p_ok() else "MISSING (a poll timed out: numbers void)")
                                    ) if _cm.p2p_active() else "all-reduce through the process group"
        comm["syncbn_streams"] = "two branch streams, exchanges in place" if K.P2P_DIRECT else "hub stream + lockstep issue"
    exposed = None
    if use_dist and not args.no_comm_ablation:
        # exposed communication = this step minus the SAME step without any collective: SyncBatchNorm modules on local
        # statistics (K.SYNC_OFF) and no gradient reducer, captured and timed the same way.  (Ranks drift apart from here
        # on -- nothing after this uses the parameters.)
        K.SYNC_OFF = True
        quiet = TrainStep(net, crit_pose, crit_par, opt, reducer=None, graph=use_graph, warmup=2)
        if reducer is not None:
            reducer.remove()
        for _ in range((3 if use_graph else 1) + args.warmup):
            quiet(images, lpar, lpose)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            quiet(images, lpar, lpose)
        barrier()
        tq = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(tq, op=dist.ReduceOp.MAX)
        if quiet.graphed == train_step.graphed:      # (a capture that fell back to eager would time the host instead)
            exposed = round((elapsed - float(tq)) / args.steps * 1e3, 3)
        K.SYNC_OFF = False
    imgs = args.batch * world * args.steps
    value = imgs / elapsed
    out = {
        "metric": "img/s fwd+bwd, model_augment @ %dx%d; 1/2/4/8 MI355X scaling" % (args.size, args.size),
        "value": round(value, 3), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": ("model_augment.Network fixed genotype C=64 L=16 R=1" if args.model == "augment" else
                                "model_search_interact.Network supernet C=32 L=16 " +
                                ("(train_with_alpha iteration: weights pass + alpha pass on a second batch)" if args.alpha_pass
                                 else "(weights pass)")) +
                               ", %dx%d, batch %d/GPU, fwd + Criterion_par + Criterion_pose + bwd + Adam step"
                               % (args.size, args.size, args.batch),
                   "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default"), "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                   "sync_bn": bool(sync_bn), "hip_graph": graph is not None or (args.launcher == "auto" and net._auto is not None and net._auto.graph is not None),
                   "launcher_loop": args.launcher or None, "loss": float(loss.detach())},
        "model_tflops": round(value * 3 * (FWD_GFLOP_PER_IMG.get(args.size, 243.37 * (args.size / 384.0) ** 2)
                                           if args.model == "augment" else 88.70 * (args.size / 384.0) ** 2) / 1e3, 2),
    }
    if comm is not None:
        out.update(comm)
        out["exposed_comm_ms"] = exposed      # ms per step; None with --no-comm-ablation
        out["ranks_bit_identical"] = ranks_identical
    if roof is not None:
        out["roofline"] = roof
        if len(roofs) > 1:
            out["roofline_other_kernels"] = roofs[1:]
    if os.environ.get("NPP_BENCH_FALLBACK"):
        out["fallback"] = os.environ["NPP_BENCH_FALLBACK"]
    if rank == 0 and os.environ.get("NPP_BENCH_RESULT"):      # the record is safe from here on (supervise_rank)
        with open(os.environ["NPP_BENCH_RESULT"], "w") as fh:
            fh.write(json.dumps(out) + "\n")
    _hb("record written")
    if use_dist and dist.get_backend() == "nccl":
        # the communicator's own word: ncclCommCount of the library's RCCL communicator over the same ranks (npp_comm_world)
        try:
            from npp_amd import comm as _c
            if not _c.active():
                _c.enable()
            out["rccl_ranks"] = int(_lib.lib().npp_comm_world())
            out["rccl_ranks_source"] = "ncclCommCount of libnpp_hip's RCCL communicator over the step's ranks (npp_comm_world)"
        except Exception as e:      # noqa: BLE001
            out["rccl_ranks_source"] = (out.get("rccl_ranks_source") or str(out.get("comm", {}).get("rccl_ranks_source", ""))) + \
                f"; ncclCommCount unavailable ({type(e).__name__}: {str(e)[:120]})"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.size)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL prints a version banner through C stdio (flushed at exit, i.e. AFTER python's own prints): flush it now so
    # the JSON record is the last line of stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:      # noqa: BLE001
        pass
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
