"""Pin the CPU oracle (oracle/nppnet_oracle.py) against outputs of the real reference
(tests/golden/*.npz, produced by oracle/make_golden.py in the build container)."""
import numpy as np
import pytest
import torch

from oracle import nppnet_oracle as O
from npp_amd.synth import synth_batch, _rng
from helpers import load_golden, template_from_golden, synth_tensors, rel_err

OPS = ['none', 'avg_pool_3x3', 'max_pool_3x3', 'skip_connect', 'std_conv_3x3', 'std_conv_1x1', 'dil_conv_3x3_2',
       'dil_conv_3x3_4', 'dil_conv_5x5_4', 'se_connect', 'conv_7x1_1x7', 'sep_conv_3x3', 'sep_conv_5x5',
       'poled_conv_x1', 'poled_conv_x2']


def _op_template(g, tag):
    """state-dict template of one op from the golden's grad/ and buf/ entries (+ BN params)."""
    shapes = {}
    for k in g.files:
        if k.startswith(tag + "/grad/") or k.startswith(tag + "/buf/"):
            shapes[k.split("/", 3)[3]] = g[k].shape
    return shapes


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("name", OPS)
def test_ops_match_reference(name, stride):
    g = load_golden("ops_golden.npz")
    tag = f"{name}/s{stride}"
    C, H, N = 32, 24, 2

    class S:  # shape holder
        def __init__(self, s):
            self.shape = s
    shapes = _op_template(g, tag)
    # SE_Block's bn exists (and has buffers) even when unused; grads absent -> add BN affine by buffer names
    tmpl = {}
    for k, s in shapes.items():
        tmpl[k] = S(s)
        if k.endswith("running_mean"):
            base = k[:-len("running_mean")]
            tmpl[base + "weight"] = S(s)
            tmpl[base + "bias"] = S(s)
            tmpl[base + "running_var"] = S(s)
    t = synth_tensors(tmpl, 0, prefix=f"{name}.s{stride}.")
    for k, v in t.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    x = torch.from_numpy(_rng(f"x.{tag}").standard_normal((N, C, H, H)).astype(np.float32)).requires_grad_(True)
    c = O.Ctx(t, True)
    y = O.apply_op(c, name, "", x, stride)
    gy = torch.from_numpy(_rng(f"gy.{tag}").standard_normal(tuple(y.shape)).astype(np.float32))
    y.backward(gy)
    assert rel_err(y.detach().numpy(), g[tag + "/y"]) < 1e-5
    assert rel_err(x.grad.numpy(), g[tag + "/dx"]) < 1e-5
    for k in g.files:
        if k.startswith(tag + "/grad/"):
            pk = k.split("/", 3)[3]
            assert t[pk].grad is not None, pk
            assert rel_err(t[pk].grad.numpy(), g[k]) < 2e-5, pk
        if k.startswith(tag + "/buf/"):
            pk = k.split("/", 3)[3]
            if pk in c.new_buffers:
                assert rel_err(c.new_buffers[pk].numpy(), g[k]) < 1e-5, pk
    # eval mode with updated stats
    t2 = {k: v.detach() for k, v in t.items()}
    t2.update(c.new_buffers)
    with torch.no_grad():
        ye = O.apply_op(O.Ctx(t2, False), name, "", x.detach(), stride)
    assert rel_err(ye.numpy(), g[tag + "/y_eval"]) < 1e-5


def _run_tiny(train):
    g = load_golden("tiny_net.npz")
    C, size, n = int(g["C"]), int(g["size"]), int(g["n"])
    t = synth_tensors(template_from_golden(g), 0)
    images, lpar, lpose, _ = synth_batch(n, size, seed=0)
    return g, t, torch.from_numpy(images), [torch.from_numpy(a) for a in lpar], \
        [torch.from_numpy(a[:, :-1]) for a in lpose]


def test_tiny_net_eval_matches_reference():
    g, t, images, _, _ = _run_tiny(False)
    with torch.no_grad():
        pose_list, par_list, _ = O.network_forward(t, images, train=False)
    for i in range(2):
        assert rel_err(pose_list[i][0].numpy(), g[f"eval/pose_map{i}"]) < 1e-4
        assert rel_err(pose_list[i][1].numpy(), g[f"eval/pose_aux{i}"]) < 1e-4
        assert rel_err(par_list[i][0].numpy(), g[f"eval/par_map{i}"]) < 1e-4
        assert rel_err(par_list[i][1].numpy(), g[f"eval/edge{i}"]) < 1e-4


def test_tiny_net_train_step_matches_reference():
    g, t, images, lpar, lpose = _run_tiny(True)
    for k, v in t.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    lam_pose = torch.full((2,), -2.5, requires_grad=True)
    lam_par = torch.full((2,), 2.3, requires_grad=True)
    loss, pose_list, par_list, newb = O.train_step_loss(t, images, lpar, lpose, lam_pose, lam_par)
    loss.backward()
    for i in range(2):
        assert rel_err(pose_list[i][0].detach().numpy(), g[f"train/pose_map{i}"]) < 1e-4
        assert rel_err(par_list[i][0].detach().numpy(), g[f"train/par_map{i}"]) < 1e-4
        assert rel_err(par_list[i][1].detach().numpy(), g[f"train/edge{i}"]) < 1e-4
        assert rel_err(pose_list[i][1].detach().numpy(), g[f"train/pose_aux{i}"]) < 1e-4
    assert abs(float(loss) - float(g["train/loss"])) < 1e-4 * abs(float(g["train/loss"]))
    assert rel_err(lam_pose.grad.numpy(), g["train/grad_lamda_pose"]) < 1e-4
    assert rel_err(lam_par.grad.numpy(), g["train/grad_lamda_par"]) < 1e-4
    for k in g.files:
        if k.startswith("train/grad/"):
            pk = k[len("train/grad/"):]
            assert rel_err(t[pk].grad.numpy(), g[k]) < 5e-4, pk
        if k.startswith("train/buf/"):
            pk = k[len("train/buf/"):]
            assert rel_err(newb[pk].numpy(), g[k]) < 1e-4, pk
    # whole-model pin: every produced gradient's norm, and the set of never-produced ones
    keys = [str(s) for s in g["train/grad_norm_keys"]]
    norms = np.array([float(t[k].grad.double().norm()) for k in keys])
    assert np.abs(norms - g["train/grad_norms"]).max() / g["train/grad_norms"].max() < 1e-4
    for k in g["train/no_grad_keys"]:
        assert t[str(k)].grad is None or float(t[str(k)].grad.abs().max()) == 0.0, k


def test_criteria_match_reference():
    g = load_golden("criteria.npz")
    for name in ["small_nvalid_lt_minkept", "kth_dominates", "thresh_dominates", "confident"]:
        n, S, s, min_kept, thres, _ = g[f"par/{name}/cfg"]
        ins = {k: torch.from_numpy(g[f"par/{name}/in/{k}"]).requires_grad_(True) for k in ("par", "edge", "par2", "edge2")}
        tgt = [torch.from_numpy(g[f"par/{name}/label_par"].astype(np.int64)),
               torch.from_numpy(g[f"par/{name}/label_edge"].astype(np.int64))]
        lam = torch.tensor([2.3, 1.7], requires_grad=True)
        loss = O.criterion_par([[ins["par"], ins["edge"]], [ins["par2"], ins["edge2"]]], tgt, lam,
                               thresh=float(thres), min_kept=int(min_kept))
        loss.backward()
        assert abs(float(loss) - float(g[f"par/{name}/loss"])) < 1e-5 * abs(float(g[f"par/{name}/loss"])), name
        for k, v in ins.items():
            assert rel_err(v.grad.numpy(), g[f"par/{name}/grad/{k}"]) < 1e-5, (name, k)
        assert rel_err(lam.grad.numpy(), g[f"par/{name}/grad_lamda"]) < 1e-5
    preds = [torch.from_numpy(g[f"pose/in/{i}"]).requires_grad_(True) for i in range(4)]
    tgt = [torch.from_numpy(g["pose/target0"]), torch.from_numpy(g["pose/target1"])]
    lam = torch.tensor([-2.5, -1.0], requires_grad=True)
    loss = O.criterion_pose([[preds[0], preds[1]], [preds[2], preds[3]]], tgt, lam)
    loss.backward()
    assert abs(float(loss) - float(g["pose/loss"])) < 1e-5 * abs(float(g["pose/loss"]))
    for i in range(4):
        assert rel_err(preds[i].grad.numpy(), g[f"pose/grad/{i}"]) < 1e-5
    assert rel_err(lam.grad.numpy(), g["pose/grad_lamda"]) < 1e-5


@pytest.mark.slow
def test_full_net_matches_reference():
    g = load_golden("full_net.npz")
    t = synth_tensors(template_from_golden(g), 0)
    images, lpar, lpose, _ = synth_batch(1, 384, seed=0)
    lam_pose = torch.full((2,), -2.5)
    lam_par = torch.full((2,), 2.3)
    with torch.no_grad():
        loss, pose_list, par_list, _ = O.train_step_loss(
            t, torch.from_numpy(images), [torch.from_numpy(a) for a in lpar],
            [torch.from_numpy(a[:, :-1]) for a in lpose], lam_pose, lam_par)
    assert rel_err(pose_list[1][0].numpy(), g["train/pose_map1"]) < 1e-4
    assert rel_err(par_list[1][0].numpy(), g["train/par_map1"]) < 1e-4
    assert abs(float(loss) - float(g["train/loss"])) < 1e-4 * abs(float(g["train/loss"]))


def test_gradient_conditioning():
    """Reference point for the gradient tolerances of the GPU tests: the f32 oracle (== the reference, see above)
    against the same computation in f64.  Outputs agree to ~1e-4, but OHEM's discrete kept-pixel set makes every
    upstream gradient move by ~1e-2 under rounding-level perturbations of the logits."""
    g = load_golden("tiny_net.npz")

    def run(dt):
        t = synth_tensors(template_from_golden(g), 0, dtype=dt)
        for k, v in t.items():
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
        images, lpar, lpose, _ = synth_batch(2, 128, seed=0)
        lam_pose = torch.full((2,), -2.5, dtype=dt)
        lam_par = torch.full((2,), 2.3, dtype=dt)
        loss, pl, pr, _ = O.train_step_loss(t, torch.from_numpy(images).to(dt), [torch.from_numpy(a) for a in lpar],
                                            [torch.from_numpy(a[:, :-1]).to(dt) for a in lpose], lam_pose, lam_par)
        loss.backward()
        return t, pr
    t32, pr32 = run(torch.float32)
    t64, pr64 = run(torch.float64)
    assert rel_err(pr32[1][0].detach().numpy(), pr64[1][0].detach().numpy()) < 1e-3
    gap = rel_err(t32["stem0.0.weight"].grad.numpy(), t64["stem0.0.weight"].grad.numpy())
    assert 1e-4 < gap < 4e-2, gap


def test_search_supernet_matches_reference():
    """Config 5 (model_search_interact.Network, all 7 candidates live in 164 MixedOps): oracle vs the reference."""
    g = load_golden("search_net.npz")
    t = synth_tensors(template_from_golden(g), 0)
    arch = ["alphas1", "alphas2", "alphas3", "alphas4", "alphas_pose", "alphas_par", "betas1", "betas2", "betas3",
            "betas4", "betas_pose", "betas_par"]
    for k in arch:
        t[k] = t[k] * 8.0
    for k, v in t.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    images, lpar, lpose, _ = synth_batch(int(g["n"]), int(g["size"]), seed=0)
    pose_list, par_list, _ = O.search_network_forward(t, torch.from_numpy(images))
    lam_pose = torch.full((2,), -2.5)
    lam_par = torch.full((2,), 2.3)
    loss = (O.criterion_par(par_list, [torch.from_numpy(a) for a in lpar], lam_par).reshape(1) +
            O.criterion_pose(pose_list, [torch.from_numpy(a[:, :-1]) for a in lpose], lam_pose).reshape(1)).mean()
    loss.backward()
    for i in range(2):
        assert rel_err(pose_list[i][0].detach().numpy(), g[f"train/pose_map{i}"]) < 1e-4
        assert rel_err(par_list[i][0].detach().numpy(), g[f"train/par_map{i}"]) < 1e-4
    assert abs(float(loss) - float(g["train/loss"])) < 1e-4 * abs(float(g["train/loss"]))
    for k in g.files:
        if k.startswith("train/grad/"):
            pk = k[len("train/grad/"):]
            assert rel_err(t[pk].grad.numpy(), g[k]) < 1e-3, pk


def test_search_supernet_eval_matches_reference():
    """The supernet in eval mode (running statistics): oracle vs the reference's outputs (search_eval.npz, the pin of the bf16
    supernet check of tests/test_ops_gpu.py)."""
    g = load_golden("search_eval.npz")
    gs = load_golden("search_net.npz")
    t = synth_tensors(template_from_golden(gs), 0)
    for k in ["alphas1", "alphas2", "alphas3", "alphas4", "alphas_pose", "alphas_par", "betas1", "betas2", "betas3",
              "betas4", "betas_pose", "betas_par"]:
        t[k] = t[k] * 8.0
    images, _, _, _ = synth_batch(int(g["n"]), int(g["size"]), seed=0)
    with torch.no_grad():
        pose_list, par_list, _ = O.search_network_forward(t, torch.from_numpy(images), train=False)
    for i in range(2):
        for nm, o in (("pose_map", pose_list[i][0]), ("pose_aux", pose_list[i][1]), ("par_map", par_list[i][0]),
                      ("edge", par_list[i][1])):
            assert rel_err(o.numpy(), g[f"eval/{nm}{i}"]) < 1e-4, (nm, i)


def test_eval_parsing_tta_confusion_matches_reference():
    """validate_sync's parsing path: oracle (flip-TTA with the reference's aliased channel swap + confusion matrix) vs the
    reference's own get_confusion_matrix on the same logits (tests/golden/eval_parsing.npz)."""
    g = load_golden("eval_parsing.npz")
    pred, flip = torch.from_numpy(g["pred"]), torch.from_numpy(g["flip"])
    label = torch.from_numpy(g["label"].astype(np.int64))
    avg = O.tta_parsing_logits(pred, flip.clone(), label.shape)
    assert np.array_equal(avg.numpy()[:, :, ::4, ::4], g["avg_sub"])
    assert np.array_equal(O.confusion_matrix(label, avg, 20, 255), g["confusion"])
    up = torch.nn.functional.interpolate(pred, size=label.shape[-2:], mode="bilinear")
    assert np.array_equal(O.confusion_matrix(label, up, 20, 255), g["confusion_noflip"])
    # the aliased swap really is what the goldens hold: a true swap gives a different matrix
    b = torch.nn.functional.interpolate(flip, size=label.shape[-2:], mode="bilinear")
    t = b.clone()
    for lo, hi in ((14, 15), (16, 17), (18, 19)):
        b[:, lo], b[:, hi] = t[:, hi], t[:, lo]
    true_swap = 0.5 * (up + b.flip(3))
    assert not np.array_equal(O.confusion_matrix(label, true_swap, 20, 255), g["confusion"])


# ---- composite blocks, extended criteria, config 4 (oracle/cases.py) -----------------------------------------------------
from oracle.cases import CELL_CASES, CELL_CASES_O0, N as CELL_N, SUB, POSE_CASES, POSE_N, POSE_J, POSE_HM, CFG4_SMALL, CFG4_FULL, \
    FULL_GRAD_KEYS, FULL_GRAD_ELEMS  # noqa: E402


def _case_template(g, name):
    keys = [str(k) for k in g[f"{name}/sd_keys"]]
    shapes = [tuple(int(d) for d in str(s).split(",") if d != "") for s in g[f"{name}/sd_shapes"]]

    class S:
        def __init__(self, s):
            self.shape = s
    return {k: S(s) for k, s in zip(keys, shapes)}


def cell_inputs(name, spec):
    xs = []
    for i, shp in enumerate(spec["inputs"]):
        xs.append(None if shp is None else
                  torch.from_numpy(_rng(f"x{i}.cells.{name}").standard_normal((CELL_N,) + tuple(shp)).astype(np.float32)))
    return xs


def check_cell_case(g, name, ys, dxs, grads, bufs, tol, tol_grad, norm=rel_err, grad_floor=0.0):
    """Outputs / input gradients (stored subsampled by SUB + whole-tensor [sum, abs-max, norm]), parameter gradients and
    running statistics of one cell case against cells_golden.npz.  Returns the worst (output, gradient) errors seen."""
    worst_y = worst_g = 0.0
    bad = []
    for k, y in enumerate(ys):
        assert tuple(y.shape) == tuple(g[f"{name}/y{k}_shape"])
        st = g[f"{name}/y{k}_stats"]
        e = max(norm(y[:, :, ::SUB, ::SUB], g[f"{name}/y{k}"]), abs(np.linalg.norm(y.astype(np.float64)) - st[2]) / st[2])
        worst_y = max(worst_y, e)
        if e >= tol:
            bad.append(("y", k, e))
    for i, dx in enumerate(dxs):
        if dx is None:
            continue
        st = g[f"{name}/dx{i}_stats"]
        e = max(norm(dx[:, :, ::SUB, ::SUB], g[f"{name}/dx{i}"]), abs(np.linalg.norm(dx.astype(np.float64)) - st[2]) / st[2])
        worst_g = max(worst_g, e)
        if e >= tol_grad:
            bad.append(("dx", i, e))
    n = 0
    gkeys = [k for k in g.files if k.startswith(name + "/grad/")]
    rms = np.median([np.linalg.norm(g[k]) / np.sqrt(g[k].size) for k in gkeys])       # typical gradient element of this block
    for k in gkeys:
        if True:
            pk = k[len(name) + 6:]
            assert pk in grads and grads[pk] is not None, (name, pk)
            ref = g[k]
            if np.linalg.norm(ref) < 1e-4 * rms * np.sqrt(ref.size):
                # exact-zero gradients (a conv bias in front of BatchNorm): rounding residue on both sides
                assert np.linalg.norm(grads[pk]) < 1e-2 * rms * np.sqrt(ref.size), (name, pk)
                continue
            if grad_floor:
                # reduced-precision runs: a tensor whose gradient nearly cancels (norm far below its peers') is measured
                # against the peers' scale, not against its own residue
                e = np.linalg.norm(grads[pk].astype(np.float64) - ref) / max(np.linalg.norm(ref), grad_floor * rms * np.sqrt(ref.size))
            else:
                e = norm(grads[pk], ref)
            worst_g = max(worst_g, e)
            if e >= tol_grad:
                bad.append((pk, e))
            n += 1
    for k in g.files:
        if k.startswith(name + "/buf/") and bufs is not None:
            pk = k[len(name) + 5:]
            if pk not in bufs:
                continue       # SE_Block.bn at stride 1 is never executed (operations.py:117,126-129): statistics untouched
            e = rel_err(bufs[pk], g[k])
            if e >= max(tol, 1e-5):
                bad.append((pk, e))
    assert not bad, (name, bad)
    assert n >= 8, (name, n)
    return worst_y, worst_g


@pytest.mark.parametrize("name", list(CELL_CASES) + list(CELL_CASES_O0))
def test_cell_blocks_match_reference(name):
    g = load_golden("cells_o0_golden.npz" if name in CELL_CASES_O0 else "cells_golden.npz")
    spec = CELL_CASES_O0[name] if name in CELL_CASES_O0 else CELL_CASES[name]
    t = synth_tensors(_case_template(g, name), 0, prefix=f"cells.{name}.")
    for k, v in t.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    xs = cell_inputs(name, spec)
    for x in xs:
        if x is not None:
            x.requires_grad_(True)
    c = O.Ctx(t, True)
    ys = O.cell_case(c, "", spec, xs)
    loss = 0.
    for k, y in enumerate(ys):
        gy = torch.from_numpy(_rng(f"gy{k}.cells.{name}").standard_normal(tuple(y.shape)).astype(np.float32))
        loss = loss + (y * gy).sum()
    loss.backward()
    check_cell_case(g, name, [y.detach().numpy() for y in ys], [None if x is None else x.grad.numpy() for x in xs],
                    {k: (v.grad.numpy() if v.grad is not None else None) for k, v in t.items() if v.requires_grad},
                    {k: v.numpy() for k, v in c.new_buffers.items()}, 2e-5, 5e-5)


def pose_case_inputs(name, spec):
    r = _rng("crit2." + name)
    preds = []
    for stage in range(2):
        for (h, w) in spec["sizes"]:
            preds.append(torch.from_numpy((r.standard_normal((POSE_N, POSE_J, h, w)) * 0.3).astype(np.float32)))
    tw = torch.from_numpy(r.uniform(0.0, 1.5, (POSE_N, POSE_J, 1)).astype(np.float32))
    _, _, lpose, _ = synth_batch(POSE_N, POSE_HM * 4, seed=3)
    tgt = [torch.from_numpy(a[:, :-1].copy()) for a in lpose]
    return preds, tw, tgt


@pytest.mark.parametrize("name", list(POSE_CASES))
def test_criterion_pose_weights_and_resample_match_reference(name):
    g = load_golden("criteria2.npz")
    spec = POSE_CASES[name]
    preds, tw, tgt = pose_case_inputs(name, spec)
    for p in preds:
        p.requires_grad_(True)
    lam = torch.tensor([-2.5, -1.0], requires_grad=True)
    loss = O.criterion_pose([[preds[0], preds[1]], [preds[2], preds[3]]], tgt, lam, tw if spec["use_target_weight"] else None)
    loss.backward()
    assert rel_err(loss.detach().numpy(), g[f"{name}/loss"]) < 1e-6
    assert rel_err(lam.grad.numpy(), g[f"{name}/grad_lamda"]) < 1e-5
    for i, p in enumerate(preds):
        assert rel_err(p.grad.numpy(), g[f"{name}/grad/{i}"]) < 1e-5, i


def _cfg4_batch(s):
    from npp_amd.synth import synth_batch_hw
    images, lpar, lpose, _ = synth_batch_hw(s["n"], s["h"], s["w"], seed=0)
    return (torch.from_numpy(images), [torch.from_numpy(a) for a in lpar], [torch.from_numpy(a[:, :-1].copy()) for a in lpose])


def test_cfg4_small_matches_reference():
    """C=16 network on a 160 x 224 batch (BASELINE config 4's non-square cousin): outputs, losses, gradients."""
    g = load_golden("cfg4_net.npz")
    t = synth_tensors(template_from_golden(g), 0)
    for k, v in t.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    images, lpar, lpose = _cfg4_batch(CFG4_SMALL)
    lam_pose, lam_par = torch.full((2,), -2.5), torch.full((2,), 2.3)
    loss, pose_list, par_list, _ = O.train_step_loss(t, images, lpar, lpose, lam_pose, lam_par)
    loss.backward()
    for i in range(2):
        for nm, o in (("pose_map", pose_list[i][0]), ("pose_aux", pose_list[i][1]), ("par_map", par_list[i][0]),
                      ("edge", par_list[i][1])):
            assert rel_err(o.detach().numpy(), g[f"small/{nm}{i}"]) < 1e-4, (nm, i)
    assert rel_err(loss.detach().numpy(), g["small/loss"]) < 1e-5
    for k in g.files:
        if k.startswith("small/grad/"):
            assert rel_err(t[k[11:]].grad.numpy(), g[k]) < 2e-2, k       # (OHEM's discrete set: test_gradient_conditioning)


@pytest.mark.slow
def test_cfg4_full_512_matches_reference():
    g = load_golden("cfg4_net.npz")
    keys = [str(k) for k in g["full/sd_keys"]]
    shapes = [tuple(int(d) for d in str(s).split(",") if d != "") for s in g["full/sd_shapes"]]

    class S:
        def __init__(self, s):
            self.shape = s
    t = synth_tensors({k: S(s) for k, s in zip(keys, shapes)}, 0)
    images, lpar, lpose = _cfg4_batch(CFG4_FULL)
    with torch.no_grad():
        loss, pose_list, par_list, _ = O.train_step_loss(t, images, lpar, lpose, torch.full((2,), -2.5), torch.full((2,), 2.3))
    assert rel_err(pose_list[1][0][:, :, ::2, ::2].numpy(), g["full/pose_map1"]) < 1e-4
    assert rel_err(par_list[1][0][:, :, ::2, ::2].numpy(), g["full/par_map1"]) < 1e-4
    assert rel_err(loss.numpy(), g["full/loss"]) < 1e-5


@pytest.mark.slow
@pytest.mark.parametrize("size", [384, 512])
def test_full_net_eval_matches_reference(size):
    from npp_amd.synth import synth_batch_hw
    g = load_golden("full_net_eval.npz")
    gf = load_golden("full_net.npz")
    t = synth_tensors(template_from_golden(gf), 0)
    images, _, _, _ = synth_batch_hw(1, size, size, seed=0)
    with torch.no_grad():
        pose_list, par_list, _ = O.network_forward(t, torch.from_numpy(images), train=False)
    for i in range(2):
        for nm, o in (("pose_map", pose_list[i][0]), ("pose_aux", pose_list[i][1]), ("par_map", par_list[i][0]),
                      ("edge", par_list[i][1])):
            assert rel_err(o[:, :, ::2, ::2].numpy(), g[f"{size}/{nm}{i}"]) < 1e-4, (size, nm, i)


@pytest.mark.slow
def test_full_net_named_gradients_match_reference():
    g = load_golden("full_net_grads.npz")
    gf = load_golden("full_net.npz")
    t = synth_tensors(template_from_golden(gf), 0)
    for k in FULL_GRAD_KEYS:
        t[k].requires_grad_(True)
    images, lpar, lpose, _ = synth_batch(1, 384, seed=0)
    loss, _, _, _ = O.train_step_loss(t, torch.from_numpy(images), [torch.from_numpy(a) for a in lpar],
                                      [torch.from_numpy(a[:, :-1].copy()) for a in lpose], torch.full((2,), -2.5),
                                      torch.full((2,), 2.3))
    loss.backward()
    for k in FULL_GRAD_KEYS:
        got = t[k].grad.reshape(-1)[:FULL_GRAD_ELEMS].numpy()
        if float(g[f"norm/{k}"]) < 1e-6:
            continue           # a conv bias in front of BatchNorm (edge_layer.1.bias): exact gradient 0
        assert rel_err(got, g[f"grad/{k}"]) < 2e-2, (k, rel_err(got, g[f"grad/{k}"]))
        assert abs(float(t[k].grad.double().norm()) - float(g[f"norm/{k}"])) < 2e-2 * float(g[f"norm/{k}"]), k


def test_search_genotype_and_entropy_beta_match_reference():
    """Host-side architecture read-outs of npp_amd's supernet (no GPU involved): genotype() and entropy_beta() on the same
    non-uniform alphas / betas give exactly the reference's discrete architecture (model_search_interact.py:898-908, 913-1052)."""
    import json
    from types import SimpleNamespace as NS
    from npp_amd.model_search_interact import Network
    g = load_golden("search_extra.npz")
    gs = load_golden("search_net.npz")
    cfg = NS(DATASET=NS(NUM_CLASSES=20, NUM_JOINTS=16), SEARCH=NS(LAYERS=16, INIT_CHANNELS=16),
             MODEL=NS(DECONV_WITH_BIAS=False, HEAD='PSP', REFINE_LAYERS=1))
    net = Network(cfg)
    sd = synth_tensors(template_from_golden(gs), 0)
    for k in ["alphas1", "alphas2", "alphas3", "alphas4", "alphas_pose", "alphas_par", "betas1", "betas2", "betas3",
              "betas4", "betas_pose", "betas_par"]:
        sd[k] = sd[k] * 8.0
    net.load_state_dict(sd)
    inter, fuse = net.genotype()

    def plain(stages):
        return [[[str(n), int(i)] for n, i in st] for st in stages]
    assert [plain(t) for t in inter] == json.loads(str(g["genotype_inter"]))
    assert [plain([fuse.pose])[0], [int(i) for i in fuse.pose_concat], plain([fuse.par])[0],
            [int(i) for i in fuse.par_concat]] == json.loads(str(g["genotype_fuse"]))
    for k in g.files:
        if k.startswith("entropy_beta/"):
            n_in, steps, want = g[k]
            got = float(net.entropy_beta(int(n_in), int(steps), getattr(net, k.split("/")[1])))
            assert abs(got - want) < 1e-6, k
