"""The C-ABI boundary without a GPU: libnpp_hip.so loads, exports every function include/npp_hip.h declares, the ctypes
binding (npp_amd/_lib.py) covers exactly that set, struct layouts match, and argument validation fails loudly with
`npp_last_error()` (no kernel is launched by these calls)."""
import ctypes as C
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "npp_hip.h")
LIB = os.path.join(REPO, "npp_amd", "libnpp_hip.so")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(npp_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.isfile(LIB):
        import __graft_entry__ as g
        g.build()
    return C.CDLL(LIB)


def test_library_exports_every_declared_symbol(lib):
    names = _declared()
    assert len(names) >= 50
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_covers_exactly_the_header():
    from npp_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared()


def test_struct_layouts_match_the_header():
    from npp_amd import _lib
    assert C.sizeof(_lib.NppTensor) == 56          # void* + 5 x int64 + int32 + pad
    assert C.sizeof(_lib.NppConvGeom) == 44        # 11 x int32
    assert C.sizeof(_lib.NppAdamJob) == 64
    assert C.sizeof(_lib.NppPackJob) == 56
    assert C.sizeof(_lib.NppBnFinalizeArgs) == 88  # 8 pointers + double + int32 + 2 floats + pad


def test_bad_arguments_fail_loudly_without_a_gpu(lib):
    lib.npp_last_error.restype = C.c_char_p
    lib.npp_version.restype = C.c_char_p
    assert b"gfx950" in lib.npp_version()
    rc = lib.npp_conv_fwd(None, None, None, None, None, None, None, None)
    assert rc < 0 and b"null" in lib.npp_last_error().lower()
    rc = lib.npp_bn_finalize(None, 0, C.c_double(0), None, None, None, None, None, C.c_float(0.1), C.c_float(1e-5), None, None, 0, None)
    assert rc < 0
    lib.npp_packed_weight_elems.restype = C.c_int64
    assert lib.npp_packed_weight_elems(128, 128, 3, 3, 0) == 128 * 1152
    assert lib.npp_packed_weight_elems(20, 256, 1, 1, 0) == 32 * 256


def test_product_path_refuses_cpu_tensors():
    import torch
    from types import SimpleNamespace as NS
    from npp_amd.model_augment import Network
    cfg = NS(DATASET=NS(NUM_CLASSES=20, NUM_JOINTS=16), TRAIN=NS(LAYERS=16, INIT_CHANNELS=16),
             MODEL=NS(DECONV_WITH_BIAS=False, HEAD='PSP', REFINE_LAYERS=1))
    net = Network(cfg)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 3, 64, 64))


def test_train_step_batch_plumbing_is_shape_exact():
    """TrainStep's static-input plumbing (host logic only): the reference's batch tuple (core/function.py:72-84) survives
    flatten -> unflatten in both pose-label forms, and the signature that decides replay-vs-eager sees shapes and dtypes."""
    import torch
    from npp_amd.train_step import TrainStep
    im = torch.zeros(2, 3, 8, 8)
    lpar = [torch.zeros(2, 8, 8, dtype=torch.int64), torch.ones(2, 8, 8, dtype=torch.int64)]
    lpose = [torch.zeros(2, 16, 2, 2), torch.ones(2, 16, 2, 2)]
    w = torch.ones(2, 16, 1)
    for pose, weight in ((lpose, w), (lpose[0], None), (tuple(lpose), None)):
        flat, layout = TrainStep._flatten(im, lpar, pose, weight)
        assert len(flat) == 1 + 2 + (1 if torch.is_tensor(pose) else 2) + (weight is not None)
        im2, lpar2, pose2, w2 = TrainStep._unflatten(flat, layout)
        assert im2 is im and lpar2[0] is lpar[0] and lpar2[1] is lpar[1] and (w2 is weight)
        if torch.is_tensor(pose):
            assert pose2 is pose
        else:
            assert isinstance(pose2, list) and pose2[0] is pose[0] and pose2[1] is pose[1]
    flat, layout = TrainStep._flatten(im, lpar, lpose, None)
    sig = TrainStep._signature(flat, layout)
    flat_b, layout_b = TrainStep._flatten(im[:1], [a[:1] for a in lpar], [a[:1] for a in lpose], None)
    assert TrainStep._signature(flat_b, layout_b) != sig
    assert TrainStep._signature(*TrainStep._flatten(im.clone(), lpar, lpose, None)) == sig


def test_library_carries_the_hash_of_the_sources_in_the_tree():
    """build.sh compiles the hash of csrc/*.hip, *.h and include/npp_hip.h into npp_version(); a library left over from other sources
    (and with it bench.py's `traffic`, which is keyed by that hash) is caught here."""
    from npp_amd import _lib
    assert _lib.built_source_hash() == _lib.kernel_source_hash()
