"""CPU: the C-ABI library loads and exports every symbol include/simpb_hip.h declares (no
compute calls — there is no GPU in the build container)."""
import ctypes
import os
import re

from simpb_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "simpb_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|void|const char\*)\s+(simpb_\w+)\s*\(", text)))


def test_header_symbols_exported():
    build.build_extension()
    import torch  # noqa: F401  (HIP runtime first, see simpb_amd/_lib.py)
    handle = ctypes.CDLL(build.LIB)
    names = declared_symbols()
    assert "simpb_deformable_aggregation_forward" in names and "simpb_ms_deform_attn_grouped_forward" in names
    for n in names:
        assert hasattr(handle, n), n
    assert sorted(_lib.SIGNATURES) == names
    assert handle.simpb_abi_version() == 7


def test_bad_arguments_return_einval():
    """Argument validation runs before any HIP call, so it is checkable without a device."""
    h = _lib.lib()
    null = ctypes.c_void_p(0)
    assert h.simpb_deformable_aggregation_forward(null, null, null, null, null, null, 1, 6, 10, 256, 4, 900, 13, 8, null) == 1
    one = ctypes.c_void_p(8)
    assert h.simpb_deformable_aggregation_forward(one, one, one, one, one, one, 1, 6, 10, 250, 4, 900, 13, 8, null) == 1
    assert h.simpb_ms_deform_attn_grouped_forward(one, one, one, one, one, one, one, 1, 6, 10, 8, 30, 4, 4, 10, null) == 1
