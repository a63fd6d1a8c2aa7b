"""ctypes binding of ``libbsclip_hip.so`` (C ABI declared in ``include/bsclip.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C bioscan-clip_amd/csrc`` into
``bioscan-clip_amd/lib/``.  There is no fallback: if it is missing, importing a compute op raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_uint32, c_void_p

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# BSCLIP_LIB: another build of the same ABI (A/B experiments: `make -C bioscan-clip_amd/csrc exp`); never a non-HIP path
LIB_PATH = os.environ.get("BSCLIP_LIB") or os.path.join(_PKG_ROOT, "lib", "libbsclip_hip.so")

(EPI_BF16, EPI_F32, EPI_GELU_BF16, EPI_RESID_F32, EPI_DGELU_BF16, EPI_PATCH_F32, EPI_GELU_FP8, EPI_RESID_BF16,
 EPI_PATCH_BF16) = range(9)
KPAD = 64
LORA_COLS = 8


class EpiArgs(Structure):
    """``bsclip_epi_args`` (include/bsclip.h); ``struct_size`` is filled in, the library rejects a mismatch."""
    _fields_ = [("struct_size", c_uint32), ("bias", c_void_p), ("resid", c_void_p), ("ld_resid", c_int),
                ("aux", c_void_p), ("ld_aux", c_int), ("dropout_p", c_float), ("dropout_seed", c_uint32)]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = ctypes.sizeof(EpiArgs)


class Fp8Args(Structure):
    """``bsclip_fp8_args`` (include/bsclip.h)."""
    _fields_ = [("struct_size", c_uint32), ("form", c_int), ("alpha", c_void_p), ("a_aug", c_void_p), ("ld_a_aug", c_int),
                ("b_aug", c_void_p), ("ld_b_aug", c_int)]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = ctypes.sizeof(Fp8Args)


P, I, F, L, U = c_void_p, c_int, c_float, c_int64, c_uint32

# name -> (restype, argtypes); mirrors include/bsclip.h one to one (tests/test_00_abi.py checks both directions)
SIGNATURES = {
    "bsclip_last_error": (c_char_p, []),
    "bsclip_abi_version": (I, []),
    "bsclip_gemm_bf16": (I, [P, I, P, I, P, I, I, I, I, I, POINTER(EpiArgs), P]),
    "bsclip_init_tables": (I, [P]),
    "bsclip_gemm_set_tile": (I, [I]),
    "bsclip_gemm_set_persistent_grid": (I, [I]),
    "bsclip_epi_args_size": (I, []),
    "bsclip_gemm_fp8": (I, [P, I, P, I, P, I, I, I, I, I, POINTER(EpiArgs), POINTER(Fp8Args), P]),
    "bsclip_quantize_rows_fp8": (I, [P, I, I, P, I, P, P]),
    "bsclip_lora_baug_set": (I, [P, I, I, P, P, P, P]),
    "bsclip_layernorm_fwd": (I, [P, I, I, I, I, P, P, F, P, I, P, P, I, P, P, F, U, P]),
    "bsclip_layernorm_fwd_fp8": (I, [P, I, I, I, I, P, P, F, P, I, P, I, P, P, P, F, U, P]),
    "bsclip_layernorm_bwd": (I, [P, I, I, P, P, I, I, P, I, P, I, P, P, I, P, I, P, I, F, U, F, U, I, P]),
    "bsclip_attn_fwd": (I, [P, I, I, I, I, P, F, P, I, P, I, P, F, U, P]),
    "bsclip_attn_bwd": (I, [P, I, P, I, P, I, I, I, P, F, P, I, I, P, F, U, P]),
    "bsclip_attn_bwd_lora": (I, [P, I, P, I, P, I, I, I, P, F, P, I, I, P, P, I, P, P, P, F, U, P]),
    "bsclip_split3_rows": (I, [P, I, I, I, P, I, P]),
    "bsclip_split3_weight": (I, [P, I, I, I, P, P, I, P, I, P]),
    "bsclip_gelu_split3": (I, [P, I, I, I, P, I, P, I, P, I, P]),
    "bsclip_meanpool_tokens_f32": (I, [P, I, I, I, P, P]),
    "bsclip_attn_fwd_f32": (I, [P, I, I, I, I, P, F, P, I, P, P, I, F, U, P]),
    "bsclip_dgelu_split3": (I, [P, I, P, I, I, I, P, I, P, I, P]),
    "bsclip_split3_transpose": (I, [P, I, I, I, I, I, P, P, I, P, I, P]),
    "bsclip_softmax_meanpool_bwd_f32": (I, [P, P, P, I, I, I, P, I, P]),
    "bsclip_lora_grad_f32_workspace_floats": (L, [I, I]),
    "bsclip_lora_grad_f32": (I, [P, I, P, I, I, I, P, P, P, P, P, P]),
    "bsclip_attn_bwd_f32": (I, [P, I, P, I, P, I, P, I, I, I, P, F, P, I, P, I, F, U, P]),
    "bsclip_exact_attn_set_impl": (I, [I]),
    "bsclip_im2col_patch16": (I, [P, I, P, I, I, P]),
    "bsclip_mask_to_bias": (I, [P, I, P, P]),
    "bsclip_vit_cls_rows": (I, [P, I, P, P, I, I, I, P]),
    "bsclip_bert_embed": (I, [P, P, I, I, I, P, I, P, P, P, P]),
    "bsclip_softmax_meanpool_fwd": (I, [P, I, I, I, P, P, P]),
    "bsclip_softmax_meanpool_bwd": (I, [P, P, P, I, I, I, P, I, P]),
    "bsclip_meanpool_tokens_fwd": (I, [P, I, I, I, P, I, P]),
    "bsclip_meanpool_tokens_bwd": (I, [P, I, I, I, I, P, P]),
    "bsclip_dgelu_mul": (I, [P, I, P, I, I, I, P, I, P]),
    "bsclip_l2norm_fwd": (I, [P, I, I, P, P, P]),
    "bsclip_l2norm_bwd": (I, [P, P, P, I, I, P, P]),
    "bsclip_infonce_workspace_floats": (L, [I, I]),
    "bsclip_infonce_set_impl": (I, [I]),
    "bsclip_infonce_fwd_bwd": (I, [POINTER(c_void_p), I, P, I, I, F, I, I, P, POINTER(c_void_p), P, P]),
    "bsclip_topk_ip_workspace_floats": (L, [I, I, I]),
    "bsclip_topk_ip": (I, [P, I, P, I, I, I, P, P, P, P]),
    "bsclip_comm_unique_id_bytes": (I, []),
    "bsclip_comm_unique_id": (I, [P]),
    "bsclip_comm_init": (I, [POINTER(c_void_p), P, I, I]),
    "bsclip_comm_destroy": (I, [P]),
    "bsclip_allgather_embeddings": (I, [P, P, P, L, P, P, P]),
    "bsclip_allgather_labels": (I, [P, P, P, L, P, P, P]),
    "bsclip_allreduce_grads": (I, [P, P, L, P, P, P]),
    "bsclip_kmer_tokenize": (I, [P, P, I, I, I, P, P]),
    "bsclip_augment_images": (I, [P, P, I, L, P, I, P, P]),
    "bsclip_lora_grad_workspace_floats": (L, [I]),
    "bsclip_lora_grad": (I, [P, I, P, I, I, I, P, P, P, P, P, P, P]),
    "bsclip_lora_grad_heads": (I, [P, I, I, I, I, P, P, P, P, P, P, P, P]),
    "bsclip_lora_grad_fp8": (I, [P, I, P, I, P, I, I, I, P, P, P, P, P, P, P]),
    "bsclip_colsum": (I, [P, I, I, I, I, P, P]),
    "bsclip_transpose_bf16": (I, [P, I, I, I, P, I, P]),
    "bsclip_cast_f32_bf16": (I, [P, L, P, P]),
    "bsclip_waug_set_lora_layers": (I, [P, I, I, I, P]),
    "bsclip_ln_param_grad_workspace_floats": (L, [I]),
    "bsclip_ln_param_grad": (I, [P, I, I, P, I, I, P, I, P, I, P, P, I, F, U, P, P, P, P]),
    "bsclip_embed_grad_workspace_floats": (L, [I]),
    "bsclip_embed_grad": (I, [P, P, I, I, I, I, I, P, P, P, P, P, P]),
    "bsclip_gather_cast_rows": (I, [P, I, I, I, I, I, I, P, I, P]),
    "bsclip_transpose_colsum_workspace_floats": (L, [I, I]),
    "bsclip_transpose_colsum_bf16": (I, [P, I, I, I, P, I, P, P, P]),
    "bsclip_gemm_splitk_f32": (I, [P, I, P, I, P, I, I, I, I, I, P, P]),
    "bsclip_adamw_step": (I, [P, P, P, P, L, F, F, F, F, F, I, F, P]),
    "bsclip_adamw_step_dev": (I, [P, P, P, P, L, P, F, F, F, F, F, P]),
    "bsclip_set_dropout_step": (I, [P]),
    "bsclip_counter_add": (I, [P, U, P]),
    "bsclip_count_nonfinite": (I, [P, L, I, P, P]),
    "bsclip_clock_probe": (I, [P, P]),
}

# only in libbsclip_hip_diag.so (`make -C bioscan-clip_amd/csrc diag`, -DBSCLIP_DIAG); used by tools/, never by the product
DIAG_SIGNATURES = {
    "bsclip_gemm_diag_ablate": (I, [I]),
    "bsclip_gemm_diag": (I, [P, I, P, I, P, I, I, I, I, I, POINTER(EpiArgs), P, P]),
    "bsclip_gemm_duo_diag": (I, [P, I, P, I, P, I, I, I, I, I, POINTER(EpiArgs), P, P]),
    "bsclip_gemm_pers_diag": (I, [P, I, P, I, P, I, I, I, I, I, POINTER(EpiArgs), P, I, P]),
    "bsclip_attn_bwd_diag": (I, [P, I, P, I, P, I, I, I, F, P, I, P, P]),
    "bsclip_attn_fwd2": (I, [P, I, I, I, I, P, F, P, P, I, P, F, U, P]),
    "bsclip_attn_bwd2": (I, [P, I, P, I, P, P, I, P, I, I, I, P, F, P, I, F, U, P]),
    "bsclip_attn_bwd2_diag": (I, [P, I, P, I, P, P, I, P, I, I, I, F, P, I, P, P]),
    "bsclip_attn_bwd_pers_diag": (I, [P, I, P, I, P, P, I, P, I, I, I, F, P, I, P, P]),
}
DIAG_LIB_PATH = os.path.join(_PKG_ROOT, "lib", "libbsclip_hip_diag.so")

_lib = None
_diag_lib = None


def load():
    """Load (once) and return the ctypes handle.  Raises if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C bioscan-clip_amd/csrc`.  bioscanclip has no non-HIP compute path.")
    lib = ctypes.CDLL(LIB_PATH)
    _bind(lib, SIGNATURES)
    if lib.bsclip_epi_args_size() != ctypes.sizeof(EpiArgs):
        raise RuntimeError(f"bsclip_epi_args is {lib.bsclip_epi_args_size()} bytes in {LIB_PATH} but "
                           f"{ctypes.sizeof(EpiArgs)} in this binding: rebuild the library")
    _lib = lib
    return lib


def _bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)  # AttributeError = ABI drift between header and library
        fn.restype = res
        fn.argtypes = args


def load_diag():
    """The diagnostic library (product entry points + the phase-stamped / ablated kernel builds).  tools/ only."""
    global _diag_lib
    if _diag_lib is None:
        if not os.path.exists(DIAG_LIB_PATH):
            raise RuntimeError(f"{DIAG_LIB_PATH} is missing: build it with `make -C bioscan-clip_amd/csrc diag`")
        _diag_lib = ctypes.CDLL(DIAG_LIB_PATH)
        _bind(_diag_lib, SIGNATURES)
        _bind(_diag_lib, DIAG_SIGNATURES)
    return _diag_lib


def last_error():
    return load().bsclip_last_error().decode()


def check(rc):
    if rc != 0:
        msg = last_error()
        if msg.startswith("Too less element"):
            raise ValueError(msg)  # reference loss_func.py:35-36
        raise RuntimeError(f"libbsclip_hip: {msg} (rc={rc})")
