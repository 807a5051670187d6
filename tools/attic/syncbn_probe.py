"""Ad-hoc probe: 2-rank SyncBN vs 1-rank local BN on one GPU at a configurable size (conditioning vs bug)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch, torch.distributed as dist, torch.multiprocessing as mp
import test_syncbn_gpu as T


def worker(rank, world, port, out, sync, n, size):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import load_golden, synth_tensors, template_from_golden
    from npp_amd import _ops as K
    from npp_amd.model_augment import Network, set_compute_dtype
    from npp_amd.synth import synth_batch
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    if os.environ.get("NPP_STAGE") == "1":
        orig = dist.all_reduce
        def staged(t, *a, **kw):
            if not t.is_cuda:
                return orig(t, *a, **kw)
            h = t.detach().cpu(); orig(h, *a, **kw); t.copy_(h)
        dist.all_reduce = staged
    g = load_golden("tiny_net.npz")
    set_compute_dtype(torch.float32)
    net = Network(T._cfg(int(g["C"])))
    net.load_state_dict(synth_tensors(template_from_golden(g), 0))
    if sync:
        net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
    net = net.to(dev).train()
    per = n // world
    images, _, _, _ = synth_batch(n, size, seed=0)
    x = torch.from_numpy(images[rank * per:(rank + 1) * per]).to(dev)
    pose_list, par_list = net(x)
    outs = T._outputs(pose_list, par_list)
    loss = sum((o.float() ** 2).sum() for o in outs)
    net.zero_grad(); loss.backward(); torch.cuda.synchronize()
    res = {}
    for k, p in net.named_parameters():
        if p.grad is None: continue
        gr = p.grad.detach().double().cpu(); dist.all_reduce(gr); res["grad/" + k] = gr.numpy()
    for i, o in enumerate(outs):
        o = o.detach().float().cpu()
        lst = [torch.zeros_like(o) for _ in range(world)]
        dist.all_gather(lst, o)
        res[f"out/{i}"] = torch.cat(lst).numpy()
    if rank == 0:
        np.savez(os.path.join(out, f"r_{int(sync)}_{world}.npz"), **res)
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    n, size = int(sys.argv[1]), int(sys.argv[2])
    out = "/tmp/syncprobe"; os.makedirs(out, exist_ok=True)
    mp.spawn(worker, args=(2, 29631, out, True, n, size), nprocs=2, join=True)
    mp.spawn(worker, args=(1, 29632, out, False, n, size), nprocs=1, join=True)
    a = np.load(os.path.join(out, "r_1_2.npz")); b = np.load(os.path.join(out, "r_0_1.npz"))
    from helpers import rel_err
    from helpers import load_golden
    g = load_golden("tiny_net.npz")
    names = ["pose_map0", "pose_aux0", "pose_map1", "pose_aux1", "par_map0", "edge0", "par_map1", "edge1"]
    for i in range(8):
        print("out", i, rel_err(a[f"out/{i}"], b[f"out/{i}"]), a[f"out/{i}"].shape,
              "sync-vs-gold", rel_err(a[f"out/{i}"], g["train/" + names[i]]) if (n, size) == (2, 64) else None,
              "local-vs-gold", rel_err(b[f"out/{i}"], g["train/" + names[i]]) if (n, size) == (2, 64) else None)
    errs = sorted(((rel_err(a[k], b[k]), k) for k in a.files if k.startswith("grad/") and np.abs(b[k]).max() > 1e-3), reverse=True)
    errs = [e for e in errs if not e[1].endswith(".1.bias")]
    print("worst grads:", errs[:4]); print("median:", errs[len(errs) // 2])
