"""Worker of test_comm_gpu.py::test_training_step_on_the_library_transport.  One rank, backend nccl (= RCCL), SyncBatchNorm
network: the same captured training steps once with torch.distributed's collectives and once with the library's own
(NPP_COMM=npp: npp_allreduce_bucket / npp_syncbn_exchange).  A world of one makes every collective an identity, so the two
trajectories agree up to the order of the float atomics (first loss identical, the second to 1e-5; after that Adam's
sign-like first steps amplify the rounding); what this proves is the plumbing (streams, capture, SyncBN hub, bucket views)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29631")

import torch
import torch.distributed as dist

from test_train_step_gpu import _batch, _cfg


def run(transport, dev, overlap=True):
    from npp_amd import _ops as K, comm
    from npp_amd.criterion import Criterion_par, Criterion_pose
    from npp_amd.ddp import GradReducer, unused_parameter_names
    from npp_amd.model_augment import Network, set_compute_dtype
    from npp_amd.optim import FusedAdam
    from npp_amd.train_step import TrainStep
    os.environ["NPP_COMM"] = transport
    if transport != "npp":
        comm.disable()
    K._SYNC_EVEN_ALONE = True
    set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(Network(_cfg(8))).to(dev).train()
    cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
    opt = FusedAdam(list(net.parameters()) + list(cp.parameters()) + list(cq.parameters()), lr=1e-3)
    red = GradReducer(net, skip=unused_parameter_names(net), always_reduce=True, bucket_mb=0.5, overlap=overlap)
    assert red._npp == (transport == "npp") and comm.active() == (transport == "npp")
    step = TrainStep(net, cp, cq, opt, reducer=red, graph=True, warmup=1)
    batch = _batch(2, 96, 5, dev)
    losses = []
    for it in range(4):
        losses.append(float(step(*batch)))
        if os.environ.get("NPP_WORKER_DIAG"):
            torch.cuda.synchronize()
            nb_p = sum(int(not torch.isfinite(p).all()) for p in net.parameters())
            nb_g = sum(int(p.grad is not None and not torch.isfinite(p.grad).all()) for p in net.parameters())
            nb_b = sum(int(not torch.isfinite(b.float()).all()) for b in net.buffers())
            nb_s = sum(int(not torch.isfinite(v).all()) for st in opt.state.values() for v in st.values() if torch.is_tensor(v) and v.is_floating_point())
            nb_f = [int(not torch.isfinite(b.flat).all()) for b in red.buckets]
            if nb_s:
                names = {id(p): n for n, p in list(net.named_parameters()) + [("criterion_pose." + n, p) for n, p in cp.named_parameters()]
                         + [("criterion_par." + n, p) for n, p in cq.named_parameters()]}
                print("diag bad adam state:", [(names.get(id(p), "?"), k, tuple(v.shape), v.flatten()[:4].tolist(),
                                                 None if p.grad is None else p.grad.flatten()[:4].tolist())
                                                for p, st in opt.state.items() for k, v in st.items()
                                                if torch.is_tensor(v) and v.is_floating_point() and not torch.isfinite(v).all()][:6], flush=True)
            if nb_g or nb_p:
                print("diag bad grads:", [(n, tuple(p.shape)) for n, p in net.named_parameters()
                                          if p.grad is not None and not torch.isfinite(p.grad).all()][:8], flush=True)
            print("diag", transport, overlap, "call", it, "loss", losses[-1], "bad params", nb_p, "grads", nb_g, "buffers", nb_b,
                  "adam state", nb_s, "buckets", sum(nb_f), "of", len(nb_f), "graphed", step.graph is not None, flush=True)
    assert step.graph is not None, "the step was not captured"
    torch.cuda.synchronize()
    params = [p.detach().clone() for p in net.parameters()]
    # a wild but finite gradient element (a block reused too early and read as another type) overflows g^2: exp_avg_sq = inf, the
    # update of that element is 0 and neither loss nor parameters show it -- so the optimizer state is part of the check
    names = {id(p): n for n, p in net.named_parameters()}
    bad_state = [(names.get(id(p), "?"), k) for p, st in opt.state.items() for k, v in st.items()
                 if torch.is_tensor(v) and v.is_floating_point() and not torch.isfinite(v).all()]
    assert not bad_state, ("non-finite optimizer state", transport, overlap, bad_state[:8])
    bad = [(n, tuple(p.shape)) for n, p in net.named_parameters() if not torch.isfinite(p).all()]
    if bad:
        print("non-finite parameters after 4 steps (transport %s, overlap %s):" % (transport, overlap), bad[:12], flush=True)
    red.remove()
    return losses, params


def main():
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    l_npp, p_npp = run("npp", dev)
    l_ref, p_ref = run("torch", dev)
    assert l_npp[0] == l_ref[0] and abs(l_npp[1] - l_ref[1]) < 1e-5 * abs(l_ref[1]), (l_npp, l_ref)
    assert all(abs(a - b) < 5e-2 * abs(b) for a, b in zip(l_npp, l_ref)), (l_npp, l_ref)
    assert all(l == l for l in l_npp) and l_npp[-1] < l_npp[0]
    assert all(torch.isfinite(p).all() for p in p_npp)
    # buckets launched by finish() instead of from the hooks (overlap=False): the step may then defer and batch its weight
    # gradients, which land in the bucket slots before finish() -- same trajectory up to the order of the f32 sums
    l_late, p_late = run("torch", dev, overlap=False)
    assert l_late[0] == l_ref[0] and abs(l_late[1] - l_ref[1]) < 1e-4 * abs(l_ref[1]), (l_late, l_ref)
    assert all(abs(a - b) < 5e-2 * abs(b) for a, b in zip(l_late, l_ref)), (l_late, l_ref)
    assert all(torch.isfinite(p).all() for p in p_late)
    # overlap="tail": buckets per kind, the KxK conv weights' buckets reduced under the second group of the batched weight-gradient
    # tail (TrainStep._tail_with_reducer: the second group runs on the parsing branch's stream inside the captured step)
    l_tail, p_tail = run("torch", dev, overlap="tail")
    assert l_tail[0] == l_ref[0] and abs(l_tail[1] - l_ref[1]) < 1e-4 * abs(l_ref[1]), (l_tail, l_ref)
    assert all(abs(a - b) < 5e-2 * abs(b) for a, b in zip(l_tail, l_ref)), (l_tail, l_ref)
    assert all(torch.isfinite(p).all() for p in p_tail)
    print("OK", l_npp)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
