"""CPU checks of the drop-in boundary: the C-ABI library loads and exports exactly what include/bsclip.h declares,
and the ctypes table mirrors the header (no compute calls -- there is no GPU here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "bsclip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bsclip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from bioscanclip.hip import lib
    handle = lib.load()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/bsclip.h but not exported"
    assert handle.bsclip_abi_version() == 2


def test_ctypes_table_matches_header():
    from bioscanclip.hip import lib
    assert sorted(lib.SIGNATURES) == header_functions()


def test_argument_validation_without_gpu():
    """Entry points validate on the host before touching the device, so bad calls fail cleanly even here."""
    from bioscanclip.hip import lib
    h = lib.load()
    assert h.bsclip_gemm_bf16(None, 0, None, 0, None, 0, 1, 1, 1, 0, None, None) == -1
    assert "null operand" in lib.last_error()
    assert h.bsclip_infonce_workspace_floats(256, 1) == -1
    assert h.bsclip_infonce_workspace_floats(256, 2) > 0
