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
    assert C.sizeof(_lib.NppPackJob) == 48
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
