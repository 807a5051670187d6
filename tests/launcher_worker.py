"""Replays the set-up and one training iteration of the reference's launcher (augment_lip_sync.py:187-213 and the loop body of
core/function.py:72-107) line for line, with npp_amd installed under the reference's module names -- the "drops into
augment_lip_sync.py unchanged" claim of BASELINE.json:north_star.  Run by tests/test_launcher_sequence.py in a process of its own
(it rebinds `models` / `core` in sys.modules and creates a process group).

    python launcher_worker.py cpu    construction, SyncBatchNorm conversion, _init_params, parameter groups, checkpoint loader
    python launcher_worker.py gpu    + .cuda(), DistributedDataParallel(find_unused_parameters=True), Adam + MultiStepLR, one step
    NPP_AUTO_GRAPH=1 python launcher_worker.py gpu    the same loop with the forward / backward replayed as hipGraphs (6 iterations)
"""
import os
import sys
import tempfile
from types import SimpleNamespace as NS

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import npp_amd  # noqa: E402

npp_amd.install_as_reference_modules()

# ---- the launcher's own import lines (augment_lip_sync.py:22-30) -------------------------------------------------------------
from models.model_augment import Network  # noqa: E402
from core.criterion import Criterion_pose, Criterion_par  # noqa: E402
import models.genotypes as genotypes  # noqa: E402,F401
from models.operations import OPS  # noqa: E402,F401


def main(mode):
    gpu = mode == "gpu"
    C = 8
    config = NS(DATASET=NS(NUM_CLASSES=20, NUM_JOINTS=16), MODEL=NS(DECONV_WITH_BIAS=False, HEAD='PSP', REFINE_LAYERS=1),
                TRAIN=NS(LAYERS=16, INIT_CHANNELS=C, LR=1e-3, LR_STEP=[1, 3], LR_FACTOR=0.1), PRINT_FREQ=1)
    args = NS(local_rank=0)
    if gpu:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(args.local_rank)
        torch.distributed.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1)

    # ---- augment_lip_sync.py:187-213, verbatim up to `.cuda()` being skipped in cpu mode ---------------------------------------
    criterion1 = Criterion_pose(out_len=2, use_target_weight=False)
    criterion2 = Criterion_par(out_len=2)
    if gpu:
        criterion1, criterion2 = criterion1.cuda(), criterion2.cuda()

    model = Network(config)
    model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
    model._init_params()
    param_dicts = [
        {"params": [p for n, p in model.named_parameters() if
                    (n.startswith('cells1.') or n.startswith('cells2') or n.startswith('stem')) and p.requires_grad],
         'lr': 0.2 * config.TRAIN.LR, },
        {"params": [p for n, p in model.named_parameters() if
                    not (n.startswith('cells1.') or n.startswith('cells2') or n.startswith('stem')) and p.requires_grad], },
    ]
    n_all = sum(p.numel() for p in model.parameters())
    assert sum(p.numel() for g in param_dicts for p in g["params"]) == n_all           # the two groups cover every parameter
    assert len(param_dicts[0]["params"]) > 0 and len(param_dicts[1]["params"]) > 0
    assert any(isinstance(m, nn.SyncBatchNorm) for m in model.modules())
    assert not any(type(m) is nn.BatchNorm2d for m in model.modules())                 # convert_sync_batchnorm reached them all

    tmp = tempfile.mkdtemp()
    model.load_pretrain_backbone(path=os.path.join(tmp, "no_such_encoder.pth"))        # missing file: silently keeps the init

    if not gpu:
        # checkpoint round trip through the tolerant loader with DDP-style `module.` keys, a wrong-shaped and an unknown entry
        sd = {"module." + k: v.clone() for k, v in model.state_dict().items()}
        sd["module.not_a_key"] = torch.zeros(3)
        sd["module.stem0.0.weight"] = torch.zeros(1, 2, 3)
        torch.save(sd, os.path.join(tmp, "ckpt.pth"))
        fresh = Network(config)
        keep = fresh.state_dict()["stem0.0.weight"].clone()
        fresh.load_pretrain_backbone(path=os.path.join(tmp, "ckpt.pth"))
        for k, v in model.state_dict().items():
            if k == "stem0.0.weight":
                assert torch.equal(fresh.state_dict()[k], keep)                        # shape mismatch: own tensor kept
            else:
                assert torch.equal(fresh.state_dict()[k], v), k
        print("LAUNCHER_CPU_OK", n_all)
        return

    model = model.cuda()
    model = nn.parallel.DistributedDataParallel(
        model, device_ids=[args.local_rank], output_device=args.local_rank, find_unused_parameters=True)

    optimizer = torch.optim.Adam(param_dicts, config.TRAIN.LR)
    optimizer.add_param_group({'params': criterion1.parameters(), 'lr': 0.0001})
    optimizer.add_param_group({'params': criterion2.parameters(), 'lr': 0.0001})
    lr = torch.optim.lr_scheduler.MultiStepLR(optimizer, config.TRAIN.LR_STEP, config.TRAIN.LR_FACTOR)

    # ---- two iterations of train() (core/function.py:72-107) on a synthetic LIP-shaped batch ----------------------------------
    from npp_amd.synth import synth_batch
    device = torch.device("cuda", args.local_rank)
    model.train()
    losses = []
    before = {k: v.detach().clone() for k, v in model.module.named_parameters()}
    auto = os.environ.get("NPP_AUTO_GRAPH") == "1"
    for it in range(6 if auto else 2):
        images, labels_par, labels_pose, meta = [torch.from_numpy(a) if not isinstance(a, (list, dict)) else a
                                                 for a in synth_batch(2, 64, seed=it)]
        labels_par = [torch.from_numpy(a) for a in labels_par]
        labels_pose = [torch.from_numpy(a) for a in labels_pose]
        images = images.to(device)
        labels_par[0] = labels_par[0].long().to(device)
        labels_par[1] = labels_par[1].long().to(device)
        labels_pose[0] = labels_pose[0][:, :-1, :, :].float().to(device)
        labels_pose[1] = labels_pose[1][:, :-1, :, :].float().to(device)
        output_pose, output_par = model(images)
        pose_weight = torch.from_numpy(meta['pose_weight']).to(device)
        losses_par = criterion2(output_par, labels_par)
        losses_par = torch.unsqueeze(losses_par, 0)
        losses_pose = criterion1(output_pose, labels_pose, target_weight=pose_weight)
        losses_pose = torch.unsqueeze(losses_pose, 0)
        loss = (losses_par + losses_pose).mean()
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        lr.step()
        losses.append(float(loss.item()))
    torch.cuda.synchronize()
    import math
    assert all(math.isfinite(v) for v in losses), losses
    moved = sum(int((before[k] != v).any()) for k, v in model.module.named_parameters())
    unused = sum(1 for p in model.module.parameters() if p.grad is None or (auto and not bool(p.grad.any())))
    if auto:      # the replayed step: graph captured at the third call, never-used parameters get zeros (DDP waits for every hook)
        assert model.module._auto is not None and model.module._auto.graph is not None, "forward was never captured"
    # SE_Block.bn at stride 1 (SURVEY TL;DR 7): 116 parameters never get a gradient.  Under the replayed step "unused" is counted as
    # "all-zero gradient", which since round 4 also holds for the ~14 conv biases in front of a train-mode BatchNorm (exact zeros)
    assert (116 <= unused <= 116 + 16) if auto else unused == 116, unused
    # (round 4: the ~14 conv biases that sit directly in front of a train-mode BatchNorm -- the four 1024 -> 512 / 384 layers, the
    #  heads' first convs, Pooled_Conv -- have the exact gradient zero, which the library now returns instead of the rounding
    #  residue of a sum that cancels: Adam leaves them where they are, the reference lets them random-walk on that residue; the
    #  BatchNorm removes them from the function either way)
    assert moved >= len(before) - 116 - 8 - 16, (moved, len(before))
    assert abs(optimizer.param_groups[0]["lr"] - 0.2 * config.TRAIN.LR * (0.01 if auto else 0.1)) < 1e-12   # MultiStepLR fired at step 1 (and 3)
    sd = model.state_dict()                                                            # DDP keys: `module.` prefix
    torch.save(sd, os.path.join(tmp, "ckpt.pth"))
    fresh = Network(config)
    fresh.load_pretrain_backbone(path=os.path.join(tmp, "ckpt.pth"))
    for k, v in model.module.state_dict().items():
        assert torch.equal(fresh.state_dict()[k], v.cpu()), k
    # the library the process actually ran on
    maps = open("/proc/self/maps").read()
    assert "libnpp_hip.so" in maps
    torch.distributed.destroy_process_group()
    print("LAUNCHER_GPU_OK", losses)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "cpu")
