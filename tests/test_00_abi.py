"""CPU checks of the drop-in boundary: the C-ABI library loads and exports exactly what include/bsclip.h declares,
and the ctypes table mirrors the header (no compute calls -- there is no GPU here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(diag=False):
    """Entry points declared in include/bsclip.h: the product set, or (diag=True) the -DBSCLIP_DIAG block."""
    text = open(os.path.join(ROOT, "include", "bsclip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    blocks = re.findall(r"#ifdef BSCLIP_DIAG(.*?)#endif", text, flags=re.S)
    text = "".join(blocks) if diag else re.sub(r"#ifdef BSCLIP_DIAG.*?#endif", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bsclip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from bioscanclip.hip import lib
    handle = lib.load()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/bsclip.h but not exported"
    assert handle.bsclip_abi_version() == 10


def test_ctypes_table_matches_header():
    from bioscanclip.hip import lib
    assert sorted(lib.SIGNATURES) == header_functions()
    assert sorted(lib.DIAG_SIGNATURES) == header_functions(diag=True)


def test_product_library_has_no_diagnostic_builds():
    from bioscanclip.hip import lib
    handle = lib.load()
    for n in header_functions(diag=True):
        assert not hasattr(handle, n), f"{n} is a -DBSCLIP_DIAG entry point and must not ship in libbsclip_hip.so"


def test_epi_args_struct_matches_library_and_is_checked():
    """The binding's struct is the library's (size exported), INTEGRATION.md documents the same fields, and a caller that
    hands over a struct of another size (a stale binding) is refused instead of being read past its end."""
    import ctypes
    from bioscanclip.hip import lib
    h = lib.load()
    assert h.bsclip_epi_args_size() == ctypes.sizeof(lib.EpiArgs) == 56
    fields = [f[0] for f in lib.EpiArgs._fields_]
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = doc[doc.index("class EpiArgs"):doc.index("_lib.bsclip_gemm_bf16.argtypes")]
    assert re.findall(r'\("([a-z_]+)",', stub) == fields
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "bsclip.h")).read(), flags=re.S)
    body = hdr[hdr.index("typedef struct bsclip_epi_args {"):hdr.index("} bsclip_epi_args;")]
    assert re.findall(r"(\w+);", body) == fields
    a = lib.EpiArgs()
    assert a.struct_size == 56
    one = ctypes.c_void_p(16)   # non-null, 16-byte aligned: the size check comes before any dereference or launch
    a.struct_size = 48
    assert h.bsclip_gemm_bf16(one, 64, one, 64, one, 128, 1, 128, 64, 0, ctypes.byref(a), None) == -1
    assert "struct_size" in lib.last_error()


def test_argument_validation_without_gpu():
    """Entry points validate on the host before touching the device, so bad calls fail cleanly even here."""
    from bioscanclip.hip import lib
    h = lib.load()
    assert h.bsclip_gemm_bf16(None, 0, None, 0, None, 0, 1, 1, 1, 0, None, None) == -1
    assert "null operand" in lib.last_error()
    assert h.bsclip_infonce_workspace_floats(256, 1) == -1
    assert h.bsclip_infonce_workspace_floats(256, 2) > 0


# entry points that launch nothing: queries, the error string and the process-global benchmarking switches (kernel selection for
# tests / tools / bench.py A-B runs; tests/conftest.py resets them after every GPU test)
CONTROL_ENTRY_POINTS = {"bsclip_last_error", "bsclip_abi_version", "bsclip_epi_args_size", "bsclip_gemm_set_tile",
                        "bsclip_gemm_set_persistent_grid", "bsclip_exact_attn_set_impl", "bsclip_infonce_set_impl"}


def test_product_library_ships_only_what_an_engine_launches():
    """Every compute entry point of libbsclip_hip.so is reached from the product package: its ``hip/ops.py`` wrapper is called by an
    engine / optimizer / loss / pipeline module (or the entry point is named there directly).  Kernels nothing launches belong in the
    diagnostic library (round 4's key-owner-sweep attention lived in the product library unused)."""
    import glob
    from bioscanclip.hip import lib
    pkg = os.path.join(ROOT, "bioscan-clip_amd")
    src = {f: open(f).read() for f in glob.glob(os.path.join(pkg, "**", "*.py"), recursive=True)}
    ops_src = src[os.path.join(pkg, "bioscanclip", "hip", "ops.py")]
    wrappers = {}
    for blk in re.split(r"\n(?=def )", ops_src):
        m = re.match(r"def (\w+)\(", blk)
        if m:
            wrappers[m.group(1)] = set(re.findall(r"\b(bsclip_\w+)\b", blk))
    callers = "\n".join(v for k, v in src.items() if not k.endswith(os.path.join("hip", "ops.py")) and not k.endswith(os.path.join("hip", "lib.py")))
    unused = []
    for name in lib.SIGNATURES:
        if name in CONTROL_ENTRY_POINTS or re.search(r"\b" + name + r"\b", callers):
            continue
        via = [f for f, e in wrappers.items() if name in e]
        if not any(re.search(r"\bops\." + f + r"\b", callers) or re.search(r"(?<!def )\b" + f + r"\(", ops_src) for f in via):
            unused.append(name)
    assert not unused, f"exported by the product library but launched by no product module: {unused}"
    assert CONTROL_ENTRY_POINTS <= set(lib.SIGNATURES)
