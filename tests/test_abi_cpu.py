"""CPU: the C-ABI library loads and exports every symbol include/ccst_hip.h declares (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    with open(os.path.join(ROOT, "include", "ccst_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ccst_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from ccst_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "libccst_hip.so does not export %s" % s
    assert sorted(_lib.EXPORTS) == syms, "ctypes binding and header disagree: %s" % (set(_lib.EXPORTS) ^ set(syms))
    assert _lib.load().ccst_abi_version() == 2 and len(syms) <= 84      # (92 until round 6: retired kernels no default path selected)


def test_conv_desc_layout_matches_header():
    from ccst_amd._lib import CcstConvDesc
    # 19 int32 + (pad) + int64 + 2 int32 + 2 int64 + 3 int32 + uint32, natural alignment
    assert ctypes.sizeof(CcstConvDesc) == 128
    assert CcstConvDesc.xsN.offset == 80 and CcstConvDesc.y_off.offset == 96 and CcstConvDesc.flags.offset == 124


def test_bad_arguments_return_errors_without_a_gpu():
    from ccst_amd import _lib
    lib = _lib.load()
    assert lib.ccst_conv2d_igemm_f32(None, None, None, None, None, None) == -1
    assert b"null" in lib.ccst_last_error()
    assert lib.ccst_stats_workspace_bytes(6, 512, 4096) > 0
    assert lib.ccst_conv2d_bwd_weight_splits(200704, 64, 256, 1) >= 1
