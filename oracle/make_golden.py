"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference)
on deterministic synthetic weights/inputs.  Runs only in the build container (the
reference never travels to the GPU box); the outputs are small data fixtures.

    python oracle/make_golden.py [--full]

Fixtures (all float32 unless noted; `torch_version` recorded in each file):
  ops_golden.npz     every OPS[name] at C=32, 24x24, N=2, stride 1 and 2: y, dx, param grads,
                     updated running stats
  tiny_net.npz       Network(C=16, L=16, R=1), 2x3x128x128: train-mode outputs, both losses,
                     ~24 named grads, some running stats; eval-mode outputs
  criteria.npz       Criterion_par / Criterion_pose on fixed logits incl. OHEM edge cases
  full_net.npz       (--full) Network(C=64), 1x3x384x384 train-mode: stage-1 par_map/pose_map
                     + summary statistics of all 8 outputs + losses
"""
from __future__ import annotations

import argparse
import os
import sys
from types import SimpleNamespace as NS

import numpy as np

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, REPO)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self  # core/criterion.py:192,197 call .cuda() unconditionally
torch.nn.Module.cuda = lambda self, *a, **k: self

from models import operations as ref_ops  # noqa: E402
from models.model_augment import Network as RefNetwork  # noqa: E402
from models.model_search_interact import Network as RefSearchNetwork  # noqa: E402
from core.criterion import Criterion_par, Criterion_pose  # noqa: E402

from npp_amd.synth import synth_state_dict, synth_batch, _rng  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def cfg(C, L=16, R=1):
    return NS(DATASET=NS(NUM_CLASSES=20, NUM_JOINTS=16), TRAIN=NS(LAYERS=L, INIT_CHANNELS=C),
              MODEL=NS(DECONV_WITH_BIAS=False, HEAD='PSP', REFINE_LAYERS=R))


def load_synth(module, seed=0, prefix=""):
    sd = module.state_dict()
    tmpl = {prefix + k: v for k, v in sd.items()}
    syn = synth_state_dict(tmpl, seed)
    module.load_state_dict({k: torch.from_numpy(syn[prefix + k]) for k in sd})


def f32(t):
    return t.detach().to(torch.float32).numpy().copy()


def gen_ops():
    out = {"torch_version": np.array(torch.__version__)}
    C, H, N = 32, 24, 2
    for name in ref_ops.OPS:
        for stride in (1, 2):
            tag = f"{name}/s{stride}"
            m = ref_ops.OPS[name](C, stride, True)
            load_synth(m, 0, prefix=f"{name}.s{stride}.")
            m.train()
            x = torch.from_numpy(_rng(f"x.{tag}").standard_normal((N, C, H, H)).astype(np.float32))
            x.requires_grad_(True)
            y = m(x)
            gy = torch.from_numpy(_rng(f"gy.{tag}").standard_normal(tuple(y.shape)).astype(np.float32))
            y.backward(gy)
            out[f"{tag}/y"] = f32(y)
            out[f"{tag}/dx"] = f32(x.grad)
            for k, p in m.named_parameters():
                if p.grad is not None:
                    out[f"{tag}/grad/{k}"] = f32(p.grad)
            for k, b in m.named_buffers():
                if k.endswith("running_mean") or k.endswith("running_var"):
                    out[f"{tag}/buf/{k}"] = f32(b)
            # eval-mode forward with the (updated) running statistics
            m.eval()
            with torch.no_grad():
                out[f"{tag}/y_eval"] = f32(m(x))
    np.savez_compressed(os.path.join(OUT, "ops_golden.npz"), **out)
    print("ops_golden.npz", len(out), "arrays")


def run_net(C, size, n, seed=0):
    torch.manual_seed(0)
    net = RefNetwork(cfg(C))
    load_synth(net, seed)
    images, lpar, lpose, meta = synth_batch(n, size, seed=seed)
    images = torch.from_numpy(images)
    lpar = [torch.from_numpy(a) for a in lpar]
    lpose = [torch.from_numpy(a[:, :-1]) for a in lpose]
    crit_pose = Criterion_pose(out_len=2, use_target_weight=False)
    crit_par = Criterion_par(out_len=2)
    net.train()
    pose_list, par_list = net(images)
    l_par = crit_par(par_list, lpar)
    l_pose = crit_pose(pose_list, lpose, target_weight=torch.from_numpy(meta["pose_weight"]))
    loss = (l_par.unsqueeze(0) + l_pose.unsqueeze(0)).mean()
    net.zero_grad()
    loss.backward()
    return net, pose_list, par_list, l_par, l_pose, loss, crit_pose, crit_par, images


def gen_tiny():
    C, size, n = 16, 128, 2
    net, pose_list, par_list, l_par, l_pose, loss, cpose, cpar, images = run_net(C, size, n)
    out = {"torch_version": np.array(torch.__version__), "C": np.array(C), "size": np.array(size), "n": np.array(n)}
    for i in range(2):
        out[f"train/pose_map{i}"] = f32(pose_list[i][0])
        out[f"train/pose_aux{i}"] = f32(pose_list[i][1])
        out[f"train/par_map{i}"] = f32(par_list[i][0])
        out[f"train/edge{i}"] = f32(par_list[i][1])
    out["train/loss_par"] = f32(l_par)
    out["train/loss_pose"] = f32(l_pose)
    out["train/loss"] = f32(loss)
    out["train/grad_lamda_pose"] = f32(cpose.lamda.grad)
    out["train/grad_lamda_par"] = f32(cpar.lamda.grad)
    names = [k for k, p in net.named_parameters() if p.grad is not None]
    want = ["stem0.0.weight", "stem5.1.weight", "cells1.0.preprocess0.net.1.weight", "cells1.0._ops.0.net.1.weight",
            "cells1.0._ops.1.conv1.weight", "cells1.0._ops.1.conv2.bias", "cells1.0._ops.4.bn.weight",
            "cells1.4._ops.4.net.1.weight", "cells1.4._ops.4.net.2.weight", "cells2.5.preprocess0.conv2.weight",
            "cells2.5.preprocess0.bn.bias", "cells2.15._ops.7.net.2.bias", "_ops1.2.1.1.weight", "_ops2.1.net.2.weight",
            "_ops2.1.net.2.bias", "up_ops1.2.1.1.bias", "upsamples1.0._ops.3.0.net.1.weight",
            "upsamples2.2._ops.4.net.3.weight", "pose_layer.1.weight", "pose_layer.1.bias", "edge_layer.2.weight",
            "pose_net.0.preprocess1.net.1.weight", "par_net.2._ops.1.conv1.weight", "pose_head.1.4.weight",
            "edge_head.0.1.weight", "edge_head.1.4.bias", "par_head.0.2.bias", "pose_auxnet.1.1.weight"]
    for k in want:
        assert k in names, k
        out[f"train/grad/{k}"] = f32(dict(net.named_parameters())[k].grad)
    # L2 norm of every produced gradient (cheap whole-model pin) + the list of never-produced ones
    out["train/grad_norm_keys"] = np.array(names)
    out["train/grad_norms"] = np.array([float(dict(net.named_parameters())[k].grad.double().norm()) for k in names],
                                       dtype=np.float64)
    out["train/no_grad_keys"] = np.array([k for k, p in net.named_parameters() if p.grad is None])
    sd = net.state_dict()
    out["sd_keys"] = np.array(list(sd.keys()))
    out["sd_shapes"] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
    for k in ["stem0.1.running_mean", "stem0.1.running_var", "cells1.3._ops.4.bn.running_var",
              "pose_net.2._ops.7.net.2.running_mean", "edge_head.1.2.running_var", "cells1.5.preprocess0.bn.running_mean"]:
        out[f"train/buf/{k}"] = f32(sd[k])
    # eval mode, fresh synthetic running stats (reload: training above mutated them)
    load_synth(net, 0)
    net.eval()
    with torch.no_grad():
        pose_list, par_list = net(images)
    for i in range(2):
        out[f"eval/pose_map{i}"] = f32(pose_list[i][0])
        out[f"eval/pose_aux{i}"] = f32(pose_list[i][1])
        out[f"eval/par_map{i}"] = f32(par_list[i][0])
        out[f"eval/edge{i}"] = f32(par_list[i][1])
    np.savez_compressed(os.path.join(OUT, "tiny_net.npz"), **out)
    print("tiny_net.npz", len(out), "arrays; loss", float(loss))


def gen_criteria():
    out = {"torch_version": np.array(torch.__version__)}
    cases = {
        # name: (n, label size, logits size, min_kept, thres, logit scale)
        "small_nvalid_lt_minkept": (2, 64, 16, 131072, 0.9, 3.0),   # n_valid < min_kept -> index n-1
        "kth_dominates": (2, 96, 24, 4000, 0.2, 6.0),               # k-th smallest prob > thresh
        "thresh_dominates": (1, 96, 24, 100, 0.9, 1.0),             # all probs small: thresh 0.9 wins
        "confident": (1, 64, 16, 50, 0.7, 30.0),                    # most p_gt ~ 1 > thresh
    }
    for name, (n, S, s, min_kept, thres, scale) in cases.items():
        r = _rng("crit." + name)
        par = torch.from_numpy((r.standard_normal((n, 20, s, s)) * scale).astype(np.float32)).requires_grad_(True)
        edge = torch.from_numpy((r.standard_normal((n, 2, s, s)) * scale).astype(np.float32)).requires_grad_(True)
        par2 = torch.from_numpy((r.standard_normal((n, 20, s, s)) * scale).astype(np.float32)).requires_grad_(True)
        edge2 = torch.from_numpy((r.standard_normal((n, 2, s, s)) * scale).astype(np.float32)).requires_grad_(True)
        _, lpar, _, _ = synth_batch(n, S, seed=__import__("zlib").crc32(name.encode()) % 1000)
        if name == "confident":  # make logits agree with labels on the coarse grid
            lab = torch.from_numpy(lpar[0])[:, ::S // s, ::S // s].clone()
            lab[lab == 255] = 0
            with torch.no_grad():
                par.scatter_(1, lab.unsqueeze(1), 40.0)
        lpar_t = [torch.from_numpy(a) for a in lpar]
        crit = Criterion_par(out_len=2, thres=thres, min_kept=min_kept)
        with torch.no_grad():
            crit.lamda.copy_(torch.tensor([2.3, 1.7]))
        loss = crit([[par, edge], [par2, edge2]], lpar_t)
        loss.backward()
        out[f"par/{name}/cfg"] = np.array([n, S, s, min_kept, thres, scale], dtype=np.float64)
        for k, v in (("par", par), ("edge", edge), ("par2", par2), ("edge2", edge2)):
            out[f"par/{name}/in/{k}"] = f32(v)
            out[f"par/{name}/grad/{k}"] = f32(v.grad)
        out[f"par/{name}/label_par"] = lpar[0].astype(np.uint8)
        out[f"par/{name}/label_edge"] = lpar[1].astype(np.uint8)
        out[f"par/{name}/loss"] = f32(loss)
        out[f"par/{name}/grad_lamda"] = f32(crit.lamda.grad)
    # pose
    n, hm = 2, 32
    r = _rng("crit.pose")
    _, _, lpose, _ = synth_batch(n, hm * 4, seed=3)
    preds = [torch.from_numpy((r.standard_normal((n, 16, hm, hm)) * 0.3).astype(np.float32)).requires_grad_(True)
             for _ in range(4)]
    crit = Criterion_pose(out_len=2)
    with torch.no_grad():
        crit.lamda.copy_(torch.tensor([-2.5, -1.0]))
    tgt = [torch.from_numpy(a[:, :-1]) for a in lpose]
    loss = crit([[preds[0], preds[1]], [preds[2], preds[3]]], tgt)
    loss.backward()
    for i, p in enumerate(preds):
        out[f"pose/in/{i}"] = f32(p)
        out[f"pose/grad/{i}"] = f32(p.grad)
    out["pose/target0"] = lpose[0][:, :-1]
    out["pose/target1"] = lpose[1][:, :-1]
    out["pose/loss"] = f32(loss)
    out["pose/grad_lamda"] = f32(crit.lamda.grad)
    np.savez_compressed(os.path.join(OUT, "criteria.npz"), **out)
    print("criteria.npz", len(out), "arrays")


def gen_search():
    """Search supernet (config 5), C=16, 2x3x128x128, non-uniform alpha/beta: train-mode outputs, losses, selected grads
    incl. every architecture tensor."""
    C, size, n = 16, 128, 2
    torch.manual_seed(0)
    c = cfg(C)
    c.SEARCH = NS(LAYERS=16, INIT_CHANNELS=C)
    net = RefSearchNetwork(c)
    load_synth(net, 0)
    with torch.no_grad():      # synthetic alphas/betas are N(0, 0.1): spread them so the softmaxes are non-uniform
        for a in net.arch_parameters():
            a.mul_(8.0)
    images, lpar, lpose, meta = synth_batch(n, size, seed=0)
    images = torch.from_numpy(images)
    lpar = [torch.from_numpy(a) for a in lpar]
    lpose = [torch.from_numpy(a[:, :-1]) for a in lpose]
    crit_pose = Criterion_pose(out_len=2, use_target_weight=False)
    crit_par = Criterion_par(out_len=2)
    net.train()
    pose_list, par_list = net(images)
    l_par = crit_par(par_list, lpar)
    l_pose = crit_pose(pose_list, lpose)
    loss = (l_par.unsqueeze(0) + l_pose.unsqueeze(0)).mean()
    net.zero_grad()
    loss.backward()
    out = {"torch_version": np.array(torch.__version__), "C": np.array(C), "size": np.array(size), "n": np.array(n)}
    for i in range(2):
        out[f"train/pose_map{i}"] = f32(pose_list[i][0])
        out[f"train/pose_aux{i}"] = f32(pose_list[i][1])
        out[f"train/par_map{i}"] = f32(par_list[i][0])
        out[f"train/edge{i}"] = f32(par_list[i][1])
    out["train/loss"] = f32(loss)
    params = dict(net.named_parameters())
    for k in ["alphas1", "alphas2", "alphas3", "alphas4", "alphas_pose", "alphas_par", "betas1", "betas2", "betas3",
              "betas4", "betas_pose", "betas_par", "_ops1.3.extra_conv.weight", "_ops1.0._ops.0.0.net.1.weight",
              "up_ops2.7._ops.4.0.net.2.weight", "pose_net.1._ops.5._ops.6.net.2.weight", "par_net.0._ops.2._ops.2.conv1.bias",
              "stem0.0.weight", "pose_head.1.4.weight"]:
        assert params[k].grad is not None, k
        out[f"train/grad/{k}"] = f32(params[k].grad)
    names = [k for k, p in net.named_parameters() if p.grad is not None]
    out["train/grad_norm_keys"] = np.array(names)
    out["train/grad_norms"] = np.array([float(params[k].grad.double().norm()) for k in names], dtype=np.float64)
    sd = net.state_dict()
    out["sd_keys"] = np.array(list(sd.keys()))
    out["sd_shapes"] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
    out["entropy"] = f32(net.loss_entropy())
    np.savez_compressed(os.path.join(OUT, "search_net.npz"), **out)
    print("search_net.npz", len(out), "arrays; loss", float(loss))


def gen_eval():
    """validate_sync's parsing path (core/function.py:925-943 + utils/utils.py:190-216): flip-TTA logits -> confusion
    matrix.  The TTA lines are inline in validate_sync (which needs cv2), so they are executed here as the same torch
    calls; the confusion matrix is the reference's own `get_confusion_matrix`."""
    np.int = int                          # utils/utils.py:201 uses the alias numpy 2 removed
    from utils.utils import get_confusion_matrix
    rng = np.random.default_rng(0xE7A1)
    n, c, h, H = 2, 20, 32, 128
    pred = torch.from_numpy(rng.standard_normal((n, c, h, h)).astype(np.float32) * 2.0)
    flip = torch.from_numpy(rng.standard_normal((n, c, h, h)).astype(np.float32) * 2.0)
    label = rng.integers(0, c, size=(n, H, H)).astype(np.int64)
    label[:, :6, :] = 255
    label[:, :, -5:] = 255
    size = (n, H, H)
    a = F.interpolate(input=pred, size=(size[-2], size[-1]), mode='bilinear')
    b = F.interpolate(input=flip, size=(size[-2], size[-1]), mode='bilinear')
    tmp = b
    b[:, 14, :, :] = tmp[:, 15, :, :]
    b[:, 15, :, :] = tmp[:, 14, :, :]
    b[:, 16, :, :] = tmp[:, 17, :, :]
    b[:, 17, :, :] = tmp[:, 16, :, :]
    b[:, 18, :, :] = tmp[:, 19, :, :]
    b[:, 19, :, :] = tmp[:, 18, :, :]
    b = b.flip(3)
    avg = 0.5 * (a + b)
    cm = get_confusion_matrix(torch.from_numpy(label), avg, size, c, 255)
    cm_noflip = get_confusion_matrix(torch.from_numpy(label), a, size, c, 255)
    # second-largest margin of every pixel: tests may tolerate arg-max flips only where it is at rounding level
    srt = np.sort(avg.numpy(), axis=1)
    margin = srt[:, -1] - srt[:, -2]
    out = {"torch_version": np.array(torch.__version__), "pred": pred.numpy(), "flip": flip.numpy(),
           "label": label.astype(np.uint8), "avg_sub": avg.numpy()[:, :, ::4, ::4].copy(), "confusion": cm, "confusion_noflip": cm_noflip,
           "near_ties": np.array(int((margin < 1e-5).sum()))}
    np.savez_compressed(os.path.join(OUT, "eval_parsing.npz"), **out)
    print("eval_parsing.npz; pixels", int(cm.sum()), "near ties", int((margin < 1e-5).sum()))


def gen_full():
    C, size, n = 64, 384, 1
    net, pose_list, par_list, l_par, l_pose, loss, cpose, cpar, images = run_net(C, size, n)
    out = {"torch_version": np.array(torch.__version__)}
    out["train/pose_map1"] = f32(pose_list[1][0])
    out["train/par_map1"] = f32(par_list[1][0])
    stats = {}
    for i in range(2):
        for nm, t in (("pose_map", pose_list[i][0]), ("pose_aux", pose_list[i][1]),
                      ("par_map", par_list[i][0]), ("edge", par_list[i][1])):
            t = t.detach().double()
            stats[f"{nm}{i}"] = [float(t.mean()), float(t.abs().max()), float(t.norm())]
    out["train/stat_keys"] = np.array(list(stats.keys()))
    out["train/stats"] = np.array(list(stats.values()), dtype=np.float64)
    out["train/loss_par"] = f32(l_par)
    out["train/loss_pose"] = f32(l_pose)
    out["train/loss"] = f32(loss)
    names = [k for k, p in net.named_parameters() if p.grad is not None]
    out["train/grad_norm_keys"] = np.array(names)
    out["train/grad_norms"] = np.array([float(dict(net.named_parameters())[k].grad.double().norm()) for k in names])
    out["train/no_grad_keys"] = np.array([k for k, p in net.named_parameters() if p.grad is None])
    sd = net.state_dict()
    out["sd_keys"] = np.array(list(sd.keys()))
    out["sd_shapes"] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
    np.savez_compressed(os.path.join(OUT, "full_net.npz"), **out)
    print("full_net.npz; loss", float(loss))


def _stats(t):
    t = t.detach().double()
    return np.array([float(t.sum()), float(t.abs().max()), float(t.norm())], dtype=np.float64)


def build_cell_case(spec, ref=True):
    """The module(s) of one oracle/cases.py CELL_CASES entry, from the REFERENCE's classes: (module or ModuleList of edge
    ops, edge source indices or None)."""
    from models import genotypes as G
    from models.model_augment import Cell, Upsample, PoseCell1, ParCell1
    kind = spec["kind"]
    if kind == "cell":
        return Cell(G.ENCODER, *spec["args"]), None
    if kind == "upsample":
        w = spec["which"]
        return Upsample(getattr(G.DECODER, f"upsample{w}"), getattr(G.DECODER, f"upsample_concat{w}"), *spec["args"]), None
    if kind == "pose":
        return PoseCell1(G.FUSION.pose, G.FUSION.pose_concat, *spec["args"]), None
    if kind == "par":
        return ParCell1(G.FUSION.par, G.FUSION.par_concat, *spec["args"]), None
    geno = getattr(G.INTER, f"task{spec['task']}")
    if kind == "inter":
        indices, ops = RefNetwork._compile(None, geno, spec["widths"])
    else:
        C = spec["C"]
        resolution = [1, 1 / 2, 1 / 4, 1 / 8, 1 / 4, 1 / 2, 1]
        indices, ops = RefNetwork._compile3(None, geno, resolution, [int(2 * C / r) for r in resolution])
    base = sum(len(ix) for ix in indices[:spec["stage"]])
    idx = indices[spec["stage"]]
    return torch.nn.ModuleList(ops[base:base + len(idx)]), idx


def gen_cells(cases=None, fname="cells_golden.npz"):
    from oracle.cases import CELL_CASES, N, SUB
    if cases is None:
        cases = CELL_CASES
    out = {"torch_version": np.array(torch.__version__)}
    for name, spec in cases.items():
        torch.manual_seed(0)
        m, idx = build_cell_case(spec)
        load_synth(m, 0, prefix=f"cells.{name}.")
        m.train()
        xs = []
        for i, shp in enumerate(spec["inputs"]):
            if shp is None:
                xs.append(None)
                continue
            x = torch.from_numpy(_rng(f"x{i}.cells.{name}").standard_normal((N,) + tuple(shp)).astype(np.float32))
            xs.append(x.requires_grad_(True))
        if idx is None:
            ys = m(*[x for x in xs if x is not None])
            ys = list(ys) if isinstance(ys, (tuple, list)) else [ys]
        else:
            z = 0
            for j, ind in enumerate(idx):
                z = z + m[j](xs[ind])
            ys = [z]
        loss = 0.
        for k, y in enumerate(ys):
            gy = torch.from_numpy(_rng(f"gy{k}.cells.{name}").standard_normal(tuple(y.shape)).astype(np.float32))
            loss = loss + (y * gy).sum()
        loss.backward()
        for k, y in enumerate(ys):
            out[f"{name}/y{k}"] = f32(y[:, :, ::SUB, ::SUB])
            out[f"{name}/y{k}_stats"] = _stats(y)
            out[f"{name}/y{k}_shape"] = np.array(y.shape)
        for i, x in enumerate(xs):
            if x is not None:
                out[f"{name}/dx{i}"] = f32(x.grad[:, :, ::SUB, ::SUB])
                out[f"{name}/dx{i}_stats"] = _stats(x.grad)
        for k, p in m.named_parameters():
            if p.grad is not None:
                out[f"{name}/grad/{k}"] = f32(p.grad)
        for k, b in m.named_buffers():
            if k.endswith("running_mean") or k.endswith("running_var"):
                out[f"{name}/buf/{k}"] = f32(b)
        sd = m.state_dict()
        out[f"{name}/sd_keys"] = np.array(list(sd.keys()))
        out[f"{name}/sd_shapes"] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, len(out), "arrays")


def gen_cells_o0():
    """PoseCell1 / ParCell1 with order == 0 (oracle/cases.py: CELL_CASES_O0), a file of their own."""
    from oracle.cases import CELL_CASES_O0
    gen_cells(CELL_CASES_O0, "cells_o0_golden.npz")


def gen_criteria2():
    """Criterion_pose with use_target_weight=True and with heat-maps whose size differs from the targets'
    (core/criterion.py:92-96, 103-108, 113-115)."""
    from oracle.cases import POSE_CASES, POSE_N, POSE_J, POSE_HM
    out = {"torch_version": np.array(torch.__version__)}
    _, _, lpose, _ = synth_batch(POSE_N, POSE_HM * 4, seed=3)
    tgt = [torch.from_numpy(a[:, :-1]) for a in lpose]
    for name, spec in POSE_CASES.items():
        r = _rng("crit2." + name)
        preds = []
        for stage in range(2):
            for (h, w) in spec["sizes"]:
                preds.append(torch.from_numpy((r.standard_normal((POSE_N, POSE_J, h, w)) * 0.3).astype(np.float32)).requires_grad_(True))
        tw = torch.from_numpy(r.uniform(0.0, 1.5, (POSE_N, POSE_J, 1)).astype(np.float32))
        crit = Criterion_pose(out_len=2, use_target_weight=spec["use_target_weight"])
        with torch.no_grad():
            crit.lamda.copy_(torch.tensor([-2.5, -1.0]))
        loss = crit([[preds[0], preds[1]], [preds[2], preds[3]]], tgt, target_weight=tw)
        loss.backward()
        for i, p in enumerate(preds):
            out[f"{name}/grad/{i}"] = f32(p.grad)
        out[f"{name}/loss"] = f32(loss)
        out[f"{name}/grad_lamda"] = f32(crit.lamda.grad)
    np.savez_compressed(os.path.join(OUT, "criteria2.npz"), **out)
    print("criteria2.npz", len(out), "arrays")


def _run_net_hw(C, n, h, w, seed=0):
    from npp_amd.synth import synth_batch_hw
    torch.manual_seed(0)
    net = RefNetwork(cfg(C))
    load_synth(net, seed)
    images, lpar, lpose, meta = synth_batch_hw(n, h, w, seed=seed)
    images = torch.from_numpy(images)
    lpar = [torch.from_numpy(a) for a in lpar]
    lpose = [torch.from_numpy(a[:, :-1].copy()) for a in lpose]
    crit_pose = Criterion_pose(out_len=2, use_target_weight=False)
    crit_par = Criterion_par(out_len=2)
    net.train()
    pose_list, par_list = net(images)
    l_par = crit_par(par_list, lpar)
    l_pose = crit_pose(pose_list, lpose)
    loss = (l_par.unsqueeze(0) + l_pose.unsqueeze(0)).mean()
    return net, pose_list, par_list, l_par, l_pose, loss


def gen_cfg4():
    """BASELINE config 4: a C=16 network at 160 x 224 (neither side a multiple of 64: ragged tiles everywhere) with
    gradients, and the full C=64 network at 512 x 512 (forward + losses)."""
    from oracle.cases import CFG4_SMALL, CFG4_FULL
    out = {"torch_version": np.array(torch.__version__)}
    s = CFG4_SMALL
    net, pose_list, par_list, l_par, l_pose, loss = _run_net_hw(s["C"], s["n"], s["h"], s["w"])
    net.zero_grad()
    loss.backward()
    for i in range(2):
        out[f"small/pose_map{i}"] = f32(pose_list[i][0])
        out[f"small/pose_aux{i}"] = f32(pose_list[i][1])
        out[f"small/par_map{i}"] = f32(par_list[i][0])
        out[f"small/edge{i}"] = f32(par_list[i][1])
    out["small/loss_par"], out["small/loss_pose"], out["small/loss"] = f32(l_par), f32(l_pose), f32(loss)
    params = dict(net.named_parameters())
    names = [k for k, p in params.items() if p.grad is not None]
    out["small/grad_norm_keys"] = np.array(names)
    out["small/grad_norms"] = np.array([float(params[k].grad.double().norm()) for k in names], dtype=np.float64)
    for k in ["stem0.0.weight", "cells1.4._ops.4.net.1.weight", "cells2.15._ops.7.net.2.bias", "upsamples2.0._ops.4.net.2.weight",
              "_ops2.1.net.2.weight", "pose_net.0.preprocess1.net.1.weight", "par_head.1.4.weight", "edge_head.1.1.weight"]:
        out[f"small/grad/{k}"] = f32(params[k].grad)
    sd = net.state_dict()
    out["sd_keys"] = np.array(list(sd.keys()))
    out["sd_shapes"] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
    s = CFG4_FULL
    with torch.no_grad():
        net, pose_list, par_list, l_par, l_pose, loss = _run_net_hw(s["C"], s["n"], s["h"], s["w"])
    out["full/pose_map1"] = f32(pose_list[1][0][:, :, ::2, ::2])
    out["full/par_map1"] = f32(par_list[1][0][:, :, ::2, ::2])
    stats = {}
    for i in range(2):
        for nm, t in (("pose_map", pose_list[i][0]), ("pose_aux", pose_list[i][1]),
                      ("par_map", par_list[i][0]), ("edge", par_list[i][1])):
            stats[f"{nm}{i}"] = _stats(t)
    out["full/stat_keys"] = np.array(list(stats.keys()))
    out["full/stats"] = np.stack(list(stats.values()))
    out["full/loss_par"], out["full/loss_pose"], out["full/loss"] = f32(l_par), f32(l_pose), f32(loss)
    sd = net.state_dict()
    out["full/sd_keys"] = np.array(list(sd.keys()))
    out["full/sd_shapes"] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
    np.savez_compressed(os.path.join(OUT, "cfg4_net.npz"), **out)
    print("cfg4_net.npz", len(out), "arrays; loss small", float(out["small/loss"]), "full", float(loss))


def gen_search_extra():
    """Host-side architecture read-outs of the search supernet: genotype() (model_search_interact.py:913-1052) and entropy_beta
    (:898-908) on the non-uniform alphas/betas of search_net.npz."""
    C = 16
    torch.manual_seed(0)
    c = cfg(C)
    c.SEARCH = NS(LAYERS=16, INIT_CHANNELS=C)
    net = RefSearchNetwork(c)
    load_synth(net, 0)
    with torch.no_grad():
        for a in net.arch_parameters():
            a.mul_(8.0)
    inter, fuse = net.genotype()
    out = {"torch_version": np.array(torch.__version__)}
    import json

    def plain(stages):
        return [[[str(n), int(i)] for n, i in st] for st in stages]
    out["genotype_inter"] = np.array(json.dumps([plain(t) for t in inter]))
    out["genotype_fuse"] = np.array(json.dumps([plain([fuse.pose])[0], [int(i) for i in fuse.pose_concat], plain([fuse.par])[0],
                                                [int(i) for i in fuse.par_concat]]))
    for name, n_in, steps in (("betas1", 2, 3), ("betas3", 5, 3), ("betas_pose", 3, 4), ("betas_par", 2, 4)):
        out[f"entropy_beta/{name}"] = np.array([n_in, steps, float(net.entropy_beta(n_in, steps, getattr(net, name)))])
    np.savez_compressed(os.path.join(OUT, "search_extra.npz"), **out)
    print("search_extra.npz", str(out["genotype_fuse"])[:120])


def gen_search_eval():
    """Eval-mode (running statistics) forward of the search supernet of search_net.npz (C=16, 2x3x128x128, the same non-uniform
    alphas / betas): the pin for the supernet's bf16 mode (train-mode outputs of a randomly initialised network amplify storage
    rounding, see gen_full_eval)."""
    C, size, n = 16, 128, 2
    torch.manual_seed(0)
    c = cfg(C)
    c.SEARCH = NS(LAYERS=16, INIT_CHANNELS=C)
    net = RefSearchNetwork(c)
    load_synth(net, 0)
    with torch.no_grad():
        for a in net.arch_parameters():
            a.mul_(8.0)
    images, _, _, _ = synth_batch(n, size, seed=0)
    net.eval()
    with torch.no_grad():
        pose_list, par_list = net(torch.from_numpy(images))
    out = {"torch_version": np.array(torch.__version__), "C": np.array(C), "size": np.array(size), "n": np.array(n)}
    for i in range(2):
        out[f"eval/pose_map{i}"] = f32(pose_list[i][0])
        out[f"eval/pose_aux{i}"] = f32(pose_list[i][1])
        out[f"eval/par_map{i}"] = f32(par_list[i][0])
        out[f"eval/edge{i}"] = f32(par_list[i][1])
    np.savez_compressed(os.path.join(OUT, "search_eval.npz"), **out)
    print("search_eval.npz", len(out), "arrays")


def gen_full_eval():
    """Eval-mode (running statistics) forward of the full configuration at 384 x 384 and 512 x 512: the pin for the bf16 mode at
    full size -- in train mode this randomly initialised network amplifies ANY perturbation ~1.3x per cell (f32 rounding
    1e-7 -> 5e-5 at the heads, bf16 rounding 3e-3 -> 0.7: tools/bf16_drift.py), in eval mode it does not."""
    from npp_amd.synth import synth_batch_hw
    out = {"torch_version": np.array(torch.__version__)}
    torch.manual_seed(0)
    net = RefNetwork(cfg(64))
    load_synth(net, 0)
    net.eval()
    for size in (384, 512):
        images, _, _, _ = synth_batch_hw(1, size, size, seed=0)
        with torch.no_grad():
            pose_list, par_list = net(torch.from_numpy(images))
        for i in range(2):
            for nm, t in (("pose_map", pose_list[i][0]), ("pose_aux", pose_list[i][1]), ("par_map", par_list[i][0]),
                          ("edge", par_list[i][1])):
                out[f"{size}/{nm}{i}"] = f32(t[:, :, ::2, ::2])
                out[f"{size}/{nm}{i}_stats"] = _stats(t)
    np.savez_compressed(os.path.join(OUT, "full_net_eval.npz"), **out)
    print("full_net_eval.npz", len(out), "arrays")


def gen_full_grads():
    """Named per-tensor gradients of the full configuration (the run of full_net.npz): the first FULL_GRAD_ELEMS elements of
    each + its norm (BatchNorm has >= 144 samples per channel there)."""
    from oracle.cases import FULL_GRAD_KEYS, FULL_GRAD_ELEMS
    net, pose_list, par_list, l_par, l_pose, loss, cpose, cpar, images = run_net(64, 384, 1)
    out = {"torch_version": np.array(torch.__version__), "loss": f32(loss)}
    params = dict(net.named_parameters())
    for k in FULL_GRAD_KEYS:
        g = params[k].grad
        assert g is not None, k
        out[f"grad/{k}"] = f32(g.reshape(-1)[:FULL_GRAD_ELEMS])
        out[f"norm/{k}"] = np.array(float(g.double().norm()))
    np.savez_compressed(os.path.join(OUT, "full_net_grads.npz"), **out)
    print("full_net_grads.npz", len(out), "arrays")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    todo = a.only.split(",") if a.only else ["ops", "criteria", "tiny", "search", "search_extra", "search_eval", "eval", "cells", "cells_o0", "criteria2"] + \
        (["full", "full_grads", "full_eval", "cfg4"] if a.full else [])
    for t in todo:
        {"ops": gen_ops, "criteria": gen_criteria, "tiny": gen_tiny, "full": gen_full, "search": gen_search, "eval": gen_eval,
         "cells": gen_cells, "cells_o0": gen_cells_o0, "criteria2": gen_criteria2, "cfg4": gen_cfg4, "full_grads": gen_full_grads,
         "full_eval": gen_full_eval, "search_extra": gen_search_extra, "search_eval": gen_search_eval}[t]()
