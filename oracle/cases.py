"""Case tables shared by the golden generator (oracle/make_golden.py, runs the real reference) and the parity tests.
Test infrastructure only: pure data, no arithmetic.

Cell-level cases: the composite blocks of models/model_augment.py -- deep enough to exercise the
fan-out gradient accumulation, the two-sided BatchNorm backward and the in-place concatenation of this repo, shallow
enough (and with >= 2304 samples per BatchNorm channel) that every gradient holds to 1e-3 in f32.

  kind "cell"     : Cell(ENCODER, C_pp, C_p, C, reduction, reduction_prev)            model_augment.py:16-62
  kind "upsample" : Upsample(DECODER.upsample{k}, concat{k}, C_pp, C_p)               model_augment.py:64-106
  kind "pose"/"par": PoseCell1 / ParCell1(FUSION..., C, C, C, order=1)                model_augment.py:119-229
  kind "inter"    : the ops Network._compile(INTER.task{t}, widths) builds for one stage, summed over its edges
                    (model_augment.py:432-436, 576-599)
  kind "inter3"   : the same for Network._compile3(INTER.task{t}, resolutions, channels)   (model_augment.py:626-649)
inputs: (channels, height, width) per input tensor; batch = N.
"""
N = 4
SUB = 6            # spatial subsampling stride of the stored activations / input gradients

CELL_CASES = {
    "cell_normal": dict(kind="cell", args=(128, 128, 32, False, False), inputs=[(128, 48, 48), (128, 48, 48)]),
    "cell_reduce": dict(kind="cell", args=(128, 128, 64, True, False), inputs=[(128, 48, 48), (128, 48, 48)]),
    "cell_after_reduce": dict(kind="cell", args=(128, 256, 64, False, True), inputs=[(128, 48, 48), (256, 24, 24)]),
    "upsample1": dict(kind="upsample", which=1, args=(256, 128), inputs=[(256, 24, 24), (128, 48, 48)]),
    "upsample2": dict(kind="upsample", which=2, args=(256, 128), inputs=[(256, 24, 24), (128, 48, 48)]),
    "pose_cell": dict(kind="pose", args=(32, 32, 32, 1), inputs=[(96, 48, 48), (128, 48, 48), (128, 48, 48)]),
    "par_cell": dict(kind="par", args=(32, 32, 32, 1), inputs=[(96, 48, 48), (128, 48, 48), (128, 48, 48)]),
    # encoder-side cross-task edges of stage 2 (task1: std_conv_1x1 from tap 1 through Interpolate(1/2) + 1x1, std_conv_3x3
    # from tap 2): widths of a C=8 network, taps at 96 / 48 / 24 / 12
    "inter_enc": dict(kind="inter", task=1, stage=2, widths=[32, 64, 128, 256],
                      inputs=[None, (64, 48, 48), (128, 24, 24), None]),
    # decoder-side edges of stage 0 (task3: dil_conv_3x3_2 from feature 4 directly, from 2 through Interpolate(1.0) + 1x1,
    # from 1 through Interpolate(1/2) + 1x1): C = 16
    "inter_dec": dict(kind="inter3", task=3, stage=0, C=16,
                      inputs=[None, (64, 48, 48), (128, 24, 24), None, (128, 24, 24), None, None]),
}

# PoseCell1 / ParCell1 with order == 0 (model_augment.py:119-229; never built by the reference's Network, model_augment.py:357-363, but
# part of the class's surface): inputs at 1/4, 1/2 and full resolution, edge ops on inputs 0 / 1 followed by a bilinear x4 / x2,
# the first two states of fea1 resampled by F.interpolate's default (nearest) mode.  Own golden file: tests/golden/cells_o0_golden.npz.
CELL_CASES_O0 = {
    "pose_cell_o0": dict(kind="pose", args=(64, 64, 32, 0), inputs=[(64, 12, 12), (64, 24, 24), (32, 48, 48)]),
    "par_cell_o0": dict(kind="par", args=(64, 64, 32, 0), inputs=[(64, 12, 12), (64, 24, 24), (32, 48, 48)]),
}

# Criterion_pose beyond the launchers' use (core/criterion.py:92-96, 103-108, 113-115)
POSE_CASES = {
    "weighted": dict(use_target_weight=True, sizes=[(32, 32), (32, 32)]),
    "resampled": dict(use_target_weight=False, sizes=[(24, 24), (16, 40)]),
    "weighted_resampled": dict(use_target_weight=True, sizes=[(40, 40), (32, 32)]),
}
POSE_N, POSE_J, POSE_HM = 2, 16, 32

# named per-tensor gradients of the full configuration (C = 64, 1 x 3 x 384 x 384): first FULL_GRAD_ELEMS elements + norm
FULL_GRAD_KEYS = ["stem0.0.weight", "stem1.1.weight", "cells1.0.preprocess1.net.1.weight", "cells1.3._ops.0.net.2.weight",
                  "cells2.7._ops.5.net.1.weight", "cells1.8._ops.4.net.1.weight", "cells2.12.preprocess0.net.2.bias",
                  "cells1.15._ops.7.net.2.weight", "_ops1.5.net.1.weight", "up_ops2.3.0.net.2.weight",
                  "upsamples1.2._ops.3.0.net.1.weight", "pose_layer.1.weight", "par_layer.2.bias", "edge_layer.1.bias",
                  "pose_net.1._ops.4.net.1.weight", "par_net.2._ops.6.net.3.weight", "pose_head.1.4.weight",
                  "par_head.1.1.weight", "pose_auxnet.1.1.weight", "edge_head.1.4.bias"]
FULL_GRAD_ELEMS = 2048

# BASELINE config 4 (512 x 512) and a size that is not a multiple of 64
CFG4_SMALL = dict(C=16, n=2, h=160, w=224)
CFG4_FULL = dict(C=64, n=1, h=512, w=512)
