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
