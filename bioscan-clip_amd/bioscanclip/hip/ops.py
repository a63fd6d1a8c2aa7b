"""Tensor-level wrappers over the C ABI (one function per ``bsclip_*`` entry point).

torch is plumbing here (device memory + the current HIP stream); every function validates shapes, dtypes and
contiguity on the host before handing raw pointers to the library, because an out-of-bounds access in a
hand-written kernel can take the whole GPU down.  Nothing in this module computes with torch ops.
"""
import ctypes

import torch

from . import lib as _l
from .lib import (EPI_BF16, EPI_DGELU_BF16, EPI_F32, EPI_GELU_BF16, EPI_GELU_FP8, EPI_PATCH_BF16, EPI_PATCH_F32,  # noqa: F401
                  EPI_RESID_BF16, EPI_RESID_F32, KPAD,
                  EpiArgs, Fp8Args, check)

BF16 = torch.bfloat16
F32 = torch.float32


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _req(cond, msg):
    if not cond:
        raise ValueError(msg)


def _rowmajor(t, name):
    # messages are only formatted on failure: these checks run ~1 500 times per training step
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a GPU tensor")
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: expected a 2-D tensor with unit inner stride")
    return t.stride(0)


def gemm(a, b, out, epilogue=EPI_BF16, bias=None, resid=None, aux=None, M=None, K=None, dropout=None):
    """out = epilogue(a[:M, :K] @ b[:, :K].T).  a [M, >=K] bf16, b [N, >=K] bf16 (row strides may exceed K)."""
    lda, ldb, ldc = _rowmajor(a, "a"), _rowmajor(b, "b"), _rowmajor(out, "out")
    _req(a.dtype == BF16 and b.dtype == BF16, "gemm operands must be bf16")
    M = a.shape[0] if M is None else M
    K = min(a.shape[1], b.shape[1]) if K is None else K
    N = b.shape[0]
    _req(M <= a.shape[0] and K <= a.shape[1] and K <= b.shape[1], "gemm: M/K exceed operand shapes")
    want = F32 if epilogue in (EPI_F32, EPI_RESID_F32, EPI_PATCH_F32) else BF16
    if out.dtype != want:
        raise ValueError(f"gemm: out dtype {out.dtype} != {want}")
    if epilogue in (EPI_PATCH_F32, EPI_PATCH_BF16):
        _req(M % 196 == 0 and out.shape[0] >= M // 196 * 197 and out.shape[1] >= N, "gemm(PATCH): out too small")
    else:
        _req(out.shape[0] >= M and out.shape[1] >= N, "gemm: out too small")
    args = EpiArgs()
    if bias is not None:
        _req(bias.dtype == F32 and bias.numel() >= N and bias.is_contiguous(), "gemm: bias must be f32 [N]")
        args.bias = bias.data_ptr()
    if resid is not None:
        _req(resid.dtype == (BF16 if epilogue == EPI_RESID_BF16 else F32), "gemm: resid must be f32 (bf16 for EPI_RESID_BF16)")
        need = 197 if epilogue in (EPI_PATCH_F32, EPI_PATCH_BF16) else M
        _req(resid.shape[0] >= need and resid.shape[1] >= N, "gemm: resid too small")
        args.resid = resid.data_ptr()
        args.ld_resid = _rowmajor(resid, "resid")
    if aux is not None:  # gelu' side band: 8-bit codes (include/bsclip.h)
        _req(aux.dtype == torch.uint8 and aux.shape[0] >= M and aux.shape[1] >= N, "gemm: aux must be uint8 [M, N]")
        args.aux = aux.data_ptr()
        args.ld_aux = _rowmajor(aux, "aux")
    if dropout is not None:  # (p, seed): C = dropout(acc + bias) + resid
        _req(epilogue in (EPI_RESID_F32, EPI_RESID_BF16), "gemm: dropout is only defined for EPI_RESID_F32 / _BF16")
        args.dropout_p, args.dropout_seed = float(dropout[0]), int(dropout[1]) & 0xFFFFFFFF
    check(_l.load().bsclip_gemm_bf16(_p(a), lda, _p(b), ldb, _p(out), ldc, M, N, K, epilogue, ctypes.byref(args),
                                     _stream()))
    return out


FP8 = torch.float8_e4m3fn   # OCP e4m3 (gfx950); one byte per element
FP8_FORM = int(__import__("os").environ.get("BSCLIP_FP8_FORM", "1"))   # 1: 16x16x32 fp8 MFMA (faster here), 2: block-scaled 16x16x128


def gemm_fp8(a8, b8, out, alpha, bias, epilogue=EPI_BF16, resid=None, aux=None, a_aug=None, b_aug=None, M=None, K=None,
             dropout=None, form=None):
    """out = epilogue(alpha[n] * (a8[:M, :K] @ b8[:, :K].T + a_aug @ b_aug.T) + bias).  a8 [M, >=K], b8 [N, >=K] fp8 e4m3;
    a_aug [M, >=64] / b_aug [N, >=64] bf16 (the LoRA K-augmentation block) or both None."""
    lda, ldb, ldc = _rowmajor(a8, "a8"), _rowmajor(b8, "b8"), _rowmajor(out, "out")
    _req(a8.dtype == FP8 and b8.dtype == FP8, "gemm_fp8 operands must be float8_e4m3fn")
    M = a8.shape[0] if M is None else M
    K = min(a8.shape[1], b8.shape[1]) if K is None else K
    N = b8.shape[0]
    _req(M <= a8.shape[0] and K <= a8.shape[1] and K <= b8.shape[1], "gemm_fp8: M/K exceed operand shapes")
    want = {EPI_BF16: BF16, EPI_F32: F32, EPI_RESID_F32: F32, EPI_GELU_FP8: FP8}.get(epilogue)
    _req(want is not None and out.dtype == want, f"gemm_fp8: epilogue {epilogue} needs out dtype {want}")
    _req(out.shape[0] >= M and out.shape[1] >= N, "gemm_fp8: out too small")
    _req(alpha.dtype == F32 and alpha.numel() >= N and alpha.is_contiguous(), "gemm_fp8: alpha must be f32 [N]")
    _req(bias.dtype == F32 and bias.numel() >= N and bias.is_contiguous(), "gemm_fp8: bias must be f32 [N]")
    args = EpiArgs()
    args.bias = bias.data_ptr()
    if resid is not None:
        _req(resid.dtype == F32 and resid.shape[0] >= M and resid.shape[1] >= N, "gemm_fp8: resid must be f32 [M, N]")
        args.resid, args.ld_resid = resid.data_ptr(), _rowmajor(resid, "resid")
    if aux is not None:
        _req(aux.dtype == torch.uint8 and aux.shape[0] >= M and aux.shape[1] >= N, "gemm_fp8: aux must be uint8 [M, N]")
        args.aux, args.ld_aux = aux.data_ptr(), _rowmajor(aux, "aux")
    if dropout is not None:
        _req(epilogue == EPI_RESID_F32, "gemm_fp8: dropout is only defined for EPI_RESID_F32")
        args.dropout_p, args.dropout_seed = float(dropout[0]), int(dropout[1]) & 0xFFFFFFFF
    f8 = Fp8Args()
    f8.form = FP8_FORM if form is None else form
    f8.alpha = alpha.data_ptr()
    _req((a_aug is None) == (b_aug is None), "gemm_fp8: a_aug and b_aug go together")
    if a_aug is not None:
        _req(a_aug.dtype == BF16 and b_aug.dtype == BF16 and a_aug.shape[0] >= M and b_aug.shape[0] >= N
             and a_aug.shape[1] >= 64 and b_aug.shape[1] >= 64, "gemm_fp8: aug blocks must be bf16 [M,>=64] / [N,>=64]")
        f8.a_aug, f8.ld_a_aug = a_aug.data_ptr(), _rowmajor(a_aug, "a_aug")
        f8.b_aug, f8.ld_b_aug = b_aug.data_ptr(), _rowmajor(b_aug, "b_aug")
    check(_l.load().bsclip_gemm_fp8(_p(a8), lda, _p(b8), ldb, _p(out), ldc, M, N, K, epilogue, ctypes.byref(args),
                                    ctypes.byref(f8), _stream()))
    return out


def quantize_rows_fp8(w):
    """f32 [R, C] -> (fp8 e4m3 [R, C], f32 scale [R]) with w ~= q * scale[:, None] (row amax mapped to 448)."""
    _req(w.dtype == F32 and w.is_contiguous() and w.dim() == 2 and w.is_cuda and w.shape[1] % 4 == 0, "quantize_rows_fp8: f32 [R, C]")
    q = torch.empty(w.shape, dtype=FP8, device=w.device)
    sc = torch.empty(w.shape[0], dtype=F32, device=w.device)
    check(_l.load().bsclip_quantize_rows_fp8(_p(w), w.shape[0], w.shape[1], _p(q), w.shape[1], _p(sc), _stream()))
    return q, sc


def lora_baug_set(b_aug, H, bq, bv, alpha):
    _req(b_aug.dtype == BF16 and b_aug.shape[0] >= 3 * H and b_aug.shape[1] >= 64, "b_aug bf16 [3H, >=64]")
    _req(all(t.dtype == F32 and t.is_contiguous() and tuple(t.shape) == (H, 4) for t in (bq, bv)), "bq/bv f32 [H,4]")
    _req(alpha.dtype == F32 and alpha.is_contiguous() and alpha.numel() >= 3 * H, "alpha f32 [3H]")
    check(_l.load().bsclip_lora_baug_set(_p(b_aug), _rowmajor(b_aug, "b_aug"), H, _p(bq), _p(bv), _p(alpha), _stream()))


_tables_ready = False


def init_tables():
    """One-time device tables, synchronised, so GEMMs launched from several streams all see them."""
    global _tables_ready
    if not _tables_ready:
        check(_l.load().bsclip_init_tables(_stream()))
        torch.cuda.synchronize()
        _tables_ready = True


def set_gemm_tile(tile):
    check(_l.load().bsclip_gemm_set_tile(tile))


def set_gemm_persistent_grid(workgroups):
    """Workgroups of the persistent GEMM kernel's launch (tile 8); 0 = one per CU."""
    check(_l.load().bsclip_gemm_set_persistent_grid(workgroups))


def layernorm_fwd(x, gamma, beta, eps, y_bf16=None, y_f32=None, lora_a=None, stats=None, M=None, dropout=None, y_split3=None):
    """``y_split3`` (exact mode): bf16 [M, >= 3H] that receives the output as a split-bf16 GEMM operand [hi | lo | hi]."""
    ld_x = _rowmajor(x, "x")
    H = gamma.numel()
    M = x.shape[0] if M is None else M
    _req(x.dtype in (F32, BF16) and x.shape[1] >= H and M <= x.shape[0], "layernorm_fwd: bad x")
    _req(gamma.dtype == F32 and beta.dtype == F32 and beta.numel() == H, "layernorm_fwd: gamma/beta f32 [H]")
    ld_y = 0
    if y_bf16 is not None:
        ld_y = _rowmajor(y_bf16, "y_bf16")
        _req(y_bf16.dtype == BF16 and y_bf16.shape[0] >= M and y_bf16.shape[1] >= H + (KPAD if lora_a is not None else 0),
             "layernorm_fwd: y_bf16 too small")
    if y_f32 is not None:
        _req(y_f32.dtype == F32 and y_f32.is_contiguous() and y_f32.shape[0] >= M and y_f32.shape[1] == H,
             "layernorm_fwd: y_f32 must be contiguous f32 [M, H]")
    if lora_a is not None:
        _req(lora_a.dtype == F32 and lora_a.is_contiguous() and tuple(lora_a.shape) == (8, H), "lora_a must be f32 [8,H]")
    if stats is not None:
        _req(stats.dtype == F32 and stats.is_contiguous() and stats.numel() >= 2 * M, "stats must be f32 [M,2]")
    ld_y3 = 0
    if y_split3 is not None:
        ld_y3 = _rowmajor(y_split3, "y_split3")
        _req(y_split3.dtype == BF16 and y_split3.shape[0] >= M and y_split3.shape[1] >= 3 * H, "layernorm_fwd: y_split3 bf16 [M, >= 3H]")
    dp, ds = (0.0, 0) if dropout is None else (float(dropout[0]), int(dropout[1]) & 0xFFFFFFFF)
    check(_l.load().bsclip_layernorm_fwd(_p(x), ld_x, int(x.dtype == BF16), M, H, _p(gamma), _p(beta), float(eps),
                                         _p(y_bf16), ld_y, _p(y_f32), _p(y_split3), ld_y3, _p(lora_a), _p(stats), dp, ds, _stream()))


def layernorm_fwd_fp8(x, gamma, beta, eps, y_fp8, t_aug=None, y_f32=None, lora_a=None, stats=None, M=None, dropout=None):
    """LayerNorm whose GEMM operand comes out as fp8 e4m3 (y_fp8 [M, >=H]) with the LoRA t block in its own bf16 buffer."""
    ld_x = _rowmajor(x, "x")
    H = gamma.numel()
    M = x.shape[0] if M is None else M
    _req(x.dtype in (F32, BF16) and x.shape[1] >= H and M <= x.shape[0], "layernorm_fwd_fp8: bad x")
    _req(gamma.dtype == F32 and beta.dtype == F32 and beta.numel() == H, "layernorm_fwd_fp8: gamma/beta f32 [H]")
    _req(y_fp8.dtype == FP8 and y_fp8.shape[0] >= M and y_fp8.shape[1] >= H, "layernorm_fwd_fp8: y_fp8 [M, >=H]")
    _req((lora_a is None) == (t_aug is None), "layernorm_fwd_fp8: lora_a and t_aug go together")
    ld_t = 0
    if t_aug is not None:
        ld_t = _rowmajor(t_aug, "t_aug")
        _req(t_aug.dtype == BF16 and t_aug.shape[0] >= M and t_aug.shape[1] >= KPAD, "t_aug bf16 [M, >=64]")
        _req(lora_a.dtype == F32 and lora_a.is_contiguous() and tuple(lora_a.shape) == (8, H), "lora_a must be f32 [8,H]")
    if y_f32 is not None:
        _req(y_f32.dtype == F32 and y_f32.is_contiguous() and y_f32.shape[0] >= M and y_f32.shape[1] == H, "y_f32 f32 [M,H]")
    if stats is not None:
        _req(stats.dtype == F32 and stats.is_contiguous() and stats.numel() >= 2 * M, "stats must be f32 [M,2]")
    dp, ds = (0.0, 0) if dropout is None else (float(dropout[0]), int(dropout[1]) & 0xFFFFFFFF)
    check(_l.load().bsclip_layernorm_fwd_fp8(_p(x), ld_x, int(x.dtype == BF16), M, H, _p(gamma), _p(beta), float(eps),
                                             _p(y_fp8), _rowmajor(y_fp8, "y_fp8"), _p(t_aug), ld_t, _p(y_f32), _p(lora_a),
                                             _p(stats), dp, ds, _stream()))


def layernorm_bwd(x, stats, gamma, mode, g_resid=None, g_gemm=None, dt=None, lora_a=None, dx_f32=None, dx_bf16=None,
                  M=None, dropout=None, in_dropout=None, dx_split3=None):
    """``g_resid`` / ``dx_f32`` are the residual-gradient stream in / out: f32, or bf16 (half the bytes; the dtype of each
    tensor is passed on as ``resid_flags``).  ``dx_bf16`` is the next dX GEMM's operand (carries ``dropout``'s mask); the exact
    backward passes ``g_gemm`` and ``dx_bf16`` as f32 tensors, or takes the operand as ``dx_split3``: bf16 [M, >= 3H] = [hi | lo | hi],
    the A operand of the next split-bf16 dX GEMM (instead of ``dx_bf16``)."""
    ld_x = _rowmajor(x, "x")
    H = gamma.numel()
    M = x.shape[0] if M is None else M
    _req(x.dtype in (F32, BF16) and x.shape[1] >= H and M <= x.shape[0], "layernorm_bwd: bad x")
    _req(stats.dtype == F32 and stats.numel() >= 2 * M, "layernorm_bwd: stats")
    ld_g = ld_dxb = ld_gr = ld_dx = 0
    if g_resid is not None:
        ld_gr = _rowmajor(g_resid, "g_resid")
        _req(g_resid.dtype in (F32, BF16) and g_resid.shape[0] >= M and g_resid.shape[1] >= H, "g_resid must be f32 / bf16 [M,>=H]")
    if g_gemm is not None:
        ld_g = _rowmajor(g_gemm, "g_gemm")
        _req(g_gemm.dtype in (BF16, F32) and g_gemm.shape[0] >= M and g_gemm.shape[1] >= H, "g_gemm must be bf16 / f32 [M,>=H]")
    if dt is not None:
        _req(dt.dtype == F32 and dt.is_contiguous() and dt.numel() >= 8 * M, "dt must be f32 [M,8]")
        _req(lora_a is not None and tuple(lora_a.shape) == (8, H) and lora_a.dtype == F32 and lora_a.is_contiguous(),
             "lora_a must be f32 [8,H]")
    if dx_f32 is not None:
        ld_dx = _rowmajor(dx_f32, "dx_f32")
        _req(dx_f32.dtype in (F32, BF16) and dx_f32.shape[0] >= M and dx_f32.shape[1] >= H, "dx_f32 must be f32 / bf16 [M,>=H]")
    if dx_split3 is not None:
        _req(dx_bf16 is None and dx_split3.dtype == BF16 and dx_split3.shape[0] >= M and dx_split3.shape[1] >= 3 * H,
             "dx_split3: bf16 [M, >= 3H], instead of dx_bf16")
        dx_bf16 = dx_split3
    if dx_bf16 is not None:
        ld_dxb = _rowmajor(dx_bf16, "dx_bf16")
        _req(dx_bf16.dtype in (BF16, F32) and dx_bf16.shape[0] >= M and dx_bf16.shape[1] >= H, "dx_bf16 (bf16 / f32 operand) too small")
    check(_l.load().bsclip_layernorm_bwd(_p(x), ld_x, int(x.dtype == BF16), _p(stats), _p(gamma), M, H, _p(g_resid),
                                         ld_gr, _p(g_gemm), ld_g, _p(dt), _p(lora_a) if dt is not None else None,
                                         int(mode), _p(dx_f32), ld_dx, _p(dx_bf16), ld_dxb,
                                         0.0 if dropout is None else float(dropout[0]),
                                         0 if dropout is None else int(dropout[1]) & 0xFFFFFFFF,
                                         0.0 if in_dropout is None else float(in_dropout[0]),
                                         0 if in_dropout is None else int(in_dropout[1]) & 0xFFFFFFFF,
                                         (1 if g_resid is not None and g_resid.dtype == BF16 else 0)
                                         | (2 if dx_f32 is not None and dx_f32.dtype == BF16 else 0)
                                         | (4 if g_gemm is not None and g_gemm.dtype == F32 else 0)
                                         | (8 if dx_bf16 is not None and dx_bf16.dtype == F32 else 0)
                                         | (16 if dx_split3 is not None else 0), _stream()))


KEEP_WORDS = 16   # csrc/attn_common.h: 32-bit dropout keep words per (batch, head, query) row (2 lane halves x 8)


def _keep_bits_ok(keep_bits, B, S, heads, who):
    if keep_bits is not None:
        _req(keep_bits.dtype == torch.int32 and keep_bits.is_contiguous() and keep_bits.is_cuda
             and keep_bits.numel() >= B * heads * S * KEEP_WORDS, f"{who}: keep_bits int32 [B * heads, S, 2, {KEEP_WORDS // 2}]")


def attn_fwd(qkv, B, S, heads, scale, ctx, lse, key_bias=None, dropout=None, q_rows=0, keep_bits=None):
    """``keep_bits`` (int32 [B * heads, S, 2, 8], with dropout only): the forward leaves its keep decisions there for ``attn_bwd``."""
    ld_qkv, ld_ctx = _rowmajor(qkv, "qkv"), _rowmajor(ctx, "ctx")
    _keep_bits_ok(keep_bits, B, S, heads, "attn_fwd")
    _req(qkv.dtype == BF16 and ctx.dtype == BF16 and lse.dtype == F32, "attn_fwd dtypes")
    _req(qkv.shape[0] >= B * S and qkv.shape[1] >= 3 * heads * 64, "attn_fwd: qkv too small")
    _req(ctx.shape[0] >= B * S and ctx.shape[1] >= heads * 64 and lse.numel() >= B * heads * S, "attn_fwd: outputs too small")
    if key_bias is not None:
        _req(key_bias.dtype == F32 and key_bias.is_contiguous() and key_bias.numel() >= B * S, "key_bias f32 [B,S]")
    dp, ds = (0.0, 0) if dropout is None else (float(dropout[0]), int(dropout[1]) & 0xFFFFFFFF)
    check(_l.load().bsclip_attn_fwd(_p(qkv), ld_qkv, B, S, heads, _p(key_bias), float(scale), _p(ctx), ld_ctx, _p(lse),
                                    int(q_rows), _p(keep_bits), dp, ds, _stream()))


def attn_bwd(qkv, dctx, lse, B, S, heads, scale, dqkv, key_bias=None, dropout=None, q_rows=0, keep_bits=None, lora=None):
    """``keep_bits``: the words the forward of the SAME (seed, step) left; None = re-hash the decisions (same masks, slower).
    ``lora`` = (t_aug, lora_b, dt_partial, db_partial): also leave the LoRA gradients' partial sums (``lora_grad_heads`` reduces them):
    t_aug bf16 [>= B S, >= 8] with t in columns 0..7 (a view of the LayerNorm output's t block), lora_b f32 [2, H, 4], dt_partial f32
    [heads, 2, B S, 4], db_partial f32 [B heads, 2, 4, 64]."""
    ld_qkv, ld_ctx, ld_d = _rowmajor(qkv, "qkv"), _rowmajor(dctx, "dctx"), _rowmajor(dqkv, "dqkv")
    _keep_bits_ok(keep_bits, B, S, heads, "attn_bwd")
    _req(all(t.dtype == BF16 for t in (qkv, dctx, dqkv)) and lse.dtype == F32, "attn_bwd dtypes")
    _req(min(qkv.shape[0], dctx.shape[0], dqkv.shape[0]) >= B * S, "attn_bwd: rows")
    _req(qkv.shape[1] >= 3 * heads * 64 and dqkv.shape[1] >= 3 * heads * 64 and dctx.shape[1] >= heads * 64
         and lse.numel() >= B * heads * S, "attn_bwd: cols")
    if key_bias is not None:
        _req(key_bias.dtype == F32 and key_bias.is_contiguous() and key_bias.numel() >= B * S, "key_bias f32 [B,S]")
    dp, ds = (0.0, 0) if dropout is None else (float(dropout[0]), int(dropout[1]) & 0xFFFFFFFF)
    if lora is not None:
        t_aug, lora_b, dtp, dbp = lora
        ld_t = _rowmajor(t_aug, "t_aug")
        _req(t_aug.dtype == BF16 and t_aug.shape[0] >= B * S and t_aug.shape[1] >= 8, "attn_bwd: t_aug bf16 [>= B S, >= 8]")
        _req(lora_b.dtype == F32 and lora_b.is_contiguous() and tuple(lora_b.shape) == (2, heads * 64, 4), "attn_bwd: lora_b f32 [2,H,4]")
        _req(dtp.dtype == F32 and dtp.is_contiguous() and dtp.numel() >= heads * B * S * 8
             and dbp.dtype == F32 and dbp.is_contiguous() and dbp.numel() >= B * heads * 512, "attn_bwd: dt_partial / db_partial sizes")
        _req(dp == 0.0 or keep_bits is not None, "attn_bwd: the LoRA partials under dropout need the forward's keep_bits")
        check(_l.load().bsclip_attn_bwd_lora(_p(qkv), ld_qkv, _p(dctx), ld_ctx, _p(lse), B, S, heads, _p(key_bias), float(scale), _p(dqkv),
                                             ld_d, int(q_rows), _p(keep_bits), _p(t_aug), ld_t, _p(lora_b), _p(dtp), _p(dbp), dp, ds,
                                             _stream()))
        return
    check(_l.load().bsclip_attn_bwd(_p(qkv), ld_qkv, _p(dctx), ld_ctx, _p(lse), B, S, heads, _p(key_bias),
                                    float(scale), _p(dqkv), ld_d, int(q_rows), _p(keep_bits), dp, ds, _stream()))


def split3_rows(src, dst, M=None, K=None):
    """f32 [M, K] -> bf16 [M, 3K] = [hi | lo | hi]: the A operand of a split-bf16 GEMM."""
    M = src.shape[0] if M is None else M
    K = src.shape[1] if K is None else K
    _req(src.dtype == F32 and dst.dtype == BF16 and src.shape[0] >= M and src.shape[1] >= K and K % 4 == 0, "split3_rows: f32 [M, K]")
    _req(dst.shape[0] >= M and dst.shape[1] >= 3 * K, "split3_rows: dst bf16 [M, 3K]")
    check(_l.load().bsclip_split3_rows(_p(src), _rowmajor(src, "src"), M, K, _p(dst), _rowmajor(dst, "dst"), _stream()))
    return dst[:M, :3 * K]


def split3_weight(w, dst, lora_a=None, lora_b=None):
    """f32 [N, K] (+ LoRA: W + B A on the q / v rows) -> bf16 [N, 3K] = [hi | hi | lo]: the B operand of a split-bf16 GEMM."""
    N, K = w.shape
    _req(w.dtype == F32 and w.is_contiguous() and dst.dtype == BF16 and dst.is_contiguous() and tuple(dst.shape) == (N, 3 * K),
         "split3_weight: w f32 [N, K] -> dst bf16 [N, 3K]")
    H = 0
    if lora_a is not None:
        H = K
        _req(lora_a.dtype == F32 and lora_a.is_contiguous() and tuple(lora_a.shape) == (8, K) and N == 3 * K, "lora_a f32 [8, H], N = 3H")
        _req(lora_b is not None and lora_b.dtype == F32 and lora_b.is_contiguous() and tuple(lora_b.shape) == (2, K, 4), "lora_b f32 [2,H,4]")
    check(_l.load().bsclip_split3_weight(_p(w), K, N, K, _p(lora_a), _p(lora_b if lora_a is not None else None), H, _p(dst), 3 * K,
                                         _stream()))
    return dst


def gelu_split3(z, dst=None, codes=None, g32=None, M=None):
    """f32 pre-activation [M, N] -> exact GELU as bf16 [M, 3N] = [hi | lo | hi] (dst) and / or f32 [M, N] (g32), + the 8-bit gelu'
    codes the backward reads."""
    M = z.shape[0] if M is None else M
    N = z.shape[1]
    _req(z.dtype == F32 and N % 4 == 0 and (dst is not None or g32 is not None), "gelu_split3: z f32 [M, N], an output")
    if dst is not None:
        _req(dst.dtype == BF16 and dst.shape[0] >= M and dst.shape[1] >= 3 * N, "gelu_split3: dst bf16 [M, 3N]")
    if g32 is not None:
        _req(g32.dtype == F32 and g32.shape[0] >= M and g32.shape[1] >= N, "gelu_split3: g32 f32 [M, N]")
    if codes is not None:
        _req(codes.dtype == torch.uint8 and codes.shape[0] >= M and codes.shape[1] >= N, "gelu_split3: codes u8 [M, N]")
    check(_l.load().bsclip_gelu_split3(_p(z), _rowmajor(z, "z"), M, N, _p(dst), 0 if dst is None else _rowmajor(dst, "dst"), _p(codes),
                                       0 if codes is None else _rowmajor(codes, "codes"), _p(g32),
                                       0 if g32 is None else _rowmajor(g32, "g32"), _stream()))
    return None if dst is None else dst[:M, :3 * N]


def meanpool_tokens_f32(x, B, S, out):
    H = x.shape[1]
    _req(x.dtype == F32 and x.is_contiguous() and x.shape[0] >= B * S and out.dtype == F32 and out.is_contiguous()
         and out.shape[0] >= B and out.shape[1] == H, "meanpool_tokens_f32: x f32 [B*S,H], out f32 [B,H]")
    check(_l.load().bsclip_meanpool_tokens_f32(_p(x), B, S, H, _p(out), _stream()))


def attn_fwd_f32(qkv, B, S, heads, scale, ctx, lse, key_bias=None, dropout=None, ctx_split3=None):
    """``ctx_split3``: bf16 [>= B S, >= 3 heads 64] that receives ctx as the out-projection GEMM's split operand [hi | lo | hi]."""
    ld_qkv, ld_ctx = _rowmajor(qkv, "qkv"), _rowmajor(ctx, "ctx")
    ld_c3 = 0
    if ctx_split3 is not None:
        ld_c3 = _rowmajor(ctx_split3, "ctx_split3")
        _req(ctx_split3.dtype == BF16 and ctx_split3.shape[0] >= B * S and ctx_split3.shape[1] >= 3 * heads * 64, "attn_fwd_f32: ctx_split3")
    _req(qkv.dtype == F32 and ctx.dtype == F32 and lse.dtype == F32, "attn_fwd_f32 dtypes")
    _req(qkv.shape[0] >= B * S and qkv.shape[1] >= 3 * heads * 64 and ctx.shape[0] >= B * S and ctx.shape[1] >= heads * 64
         and lse.numel() >= B * heads * S, "attn_fwd_f32 shapes")
    if key_bias is not None:
        _req(key_bias.dtype == F32 and key_bias.is_contiguous() and key_bias.numel() >= B * S, "key_bias f32 [B,S]")
    dp, ds = (0.0, 0) if dropout is None else (float(dropout[0]), int(dropout[1]) & 0xFFFFFFFF)
    check(_l.load().bsclip_attn_fwd_f32(_p(qkv), ld_qkv, B, S, heads, _p(key_bias), float(scale), _p(ctx), ld_ctx, _p(lse),
                                        _p(ctx_split3), ld_c3, dp, ds, _stream()))


# ---- exact backward (BSCLIP_PARITY=2): f32 gradients, split operands on the dX / dW GEMMs (csrc/exact.hip) ----
def dgelu_split3(dact, z, dst=None, out32=None, M=None):
    """dact * gelu'(z), both f32 [M, N] -> bf16 [M, 3N] = [hi | lo | hi] (dst) and / or f32 (out32; may alias dact)."""
    M = z.shape[0] if M is None else M
    N = z.shape[1]
    _req(z.dtype == F32 and dact.dtype == F32 and dact.shape[0] >= M and dact.shape[1] >= N and N % 4 == 0
         and (dst is not None or out32 is not None), "dgelu_split3: dact, z f32 [M, N], an output")
    if dst is not None:
        _req(dst.dtype == BF16 and dst.shape[0] >= M and dst.shape[1] >= 3 * N, "dgelu_split3: dst bf16 [M, 3N]")
    if out32 is not None:
        _req(out32.dtype == F32 and out32.shape[0] >= M and out32.shape[1] >= N, "dgelu_split3: out32 f32 [M, N]")
    check(_l.load().bsclip_dgelu_split3(_p(dact), _rowmajor(dact, "dact"), _p(z), _rowmajor(z, "z"), M, N, _p(dst),
                                        0 if dst is None else _rowmajor(dst, "dst"), _p(out32),
                                        0 if out32 is None else _rowmajor(out32, "out32"), _stream()))
    return None if dst is None else dst[:M, :3 * N]


def split3_transpose(src, dst_flat, order, R=None, lora_a=None, lora_b=None):
    """src f32 [R, C] -> bf16 [C, 3 Rp] (Rp = R rounded up to 64; carved out of the flat bf16 buffer ``dst_flat``): row c is the
    split of column c, order 0 = [hi | lo | hi], 1 = [hi | hi | lo]; lora_a / lora_b fold W + B A first (src = the [3H, H] QKV weight)."""
    R = src.shape[0] if R is None else R
    C = src.shape[1]
    Rp = (R + 63) // 64 * 64
    _req(src.dtype == F32 and src.shape[0] >= R and dst_flat.dtype == BF16 and dst_flat.is_contiguous()
         and dst_flat.numel() >= C * 3 * Rp, "split3_transpose: src f32 [R, C], dst bf16 >= C * 3 * Rp")
    H = 0
    if lora_a is not None:
        H = C
        _req(src.is_contiguous() and R == 3 * C and lora_a.dtype == F32 and lora_a.is_contiguous() and tuple(lora_a.shape) == (8, C)
             and lora_b is not None and lora_b.dtype == F32 and lora_b.is_contiguous() and tuple(lora_b.shape) == (2, C, 4),
             "split3_transpose: LoRA fold needs src [3H, H], lora_a [8, H], lora_b [2, H, 4]")
    dst = dst_flat.view(-1)[:C * 3 * Rp].view(C, 3 * Rp)
    check(_l.load().bsclip_split3_transpose(_p(src), _rowmajor(src, "src"), R, C, Rp, int(order), _p(lora_a),
                                            _p(lora_b if lora_a is not None else None), H, _p(dst), 3 * Rp, _stream()))
    return dst


def softmax_meanpool_bwd_f32(logits, stats, d_pooled, B, S, dlogits):
    C = logits.shape[1]
    _req(logits.dtype == F32 and logits.is_contiguous() and stats.dtype == F32 and d_pooled.dtype == F32 and d_pooled.is_contiguous()
         and dlogits.dtype == F32 and dlogits.shape[0] >= B * S and dlogits.shape[1] >= C, "softmax_meanpool_bwd_f32: f32 tensors")
    check(_l.load().bsclip_softmax_meanpool_bwd_f32(_p(logits), _p(stats), _p(d_pooled), B, S, C, _p(dlogits),
                                                    _rowmajor(dlogits, "dlogits"), _stream()))


_LG32_WS = {}


def lora_grad_f32(dqkv, y, M, H, lora_a, lora_b, dA, dB):
    """dA [8, H] += ..., dB [2, H, 4] += ... from f32 dqkv [M, >= 3H] and the f32 LayerNorm output y [M, >= H]."""
    _req(dqkv.dtype == F32 and y.dtype == F32 and dqkv.shape[0] >= M and dqkv.shape[1] >= 3 * H and y.shape[0] >= M and y.shape[1] >= H,
         "lora_grad_f32 shapes")
    _req(all(t.dtype == F32 and t.is_contiguous() for t in (lora_a, lora_b, dA, dB)) and tuple(lora_a.shape) == (8, H)
         and tuple(lora_b.shape) == (2, H, 4) and tuple(dA.shape) == (8, H) and tuple(dB.shape) == (2, H, 4), "lora_grad_f32: parameters")
    key = (M, H, str(dqkv.device), torch.cuda.current_stream().cuda_stream)   # one workspace per launching stream (towers run concurrently)
    ws = _LG32_WS.get(key)
    if ws is None:
        ws = _LG32_WS[key] = torch.empty(_l.load().bsclip_lora_grad_f32_workspace_floats(M, H), dtype=F32, device=dqkv.device)
    check(_l.load().bsclip_lora_grad_f32(_p(dqkv), _rowmajor(dqkv, "dqkv"), _p(y), _rowmajor(y, "y"), M, H, _p(lora_a), _p(lora_b),
                                         _p(dA), _p(dB), _p(ws), _stream()))


def exact_attn_set_impl(impl):
    """0 = split-bf16 operands on the bf16 matrix cores (default, csrc/attn_x3.hip); 2 = f32-operand MFMA, 1 = vector-ALU kernels (exact-f32
    second implementations, for tests)."""
    check(_l.load().bsclip_exact_attn_set_impl(int(impl)))


def attn_bwd_f32(qkv, dctx, ctx, lse, B, S, heads, scale, dqkv, key_bias=None, dropout=None, dqkv_split3=None):
    """``dqkv_split3``: bf16 [>= B S, >= 9 heads 64] that receives [dq | dk | dv] as the QKV dX GEMM's split operand [hi | lo | hi]."""
    _req(all(t.dtype == F32 for t in (qkv, dctx, ctx, lse, dqkv)), "attn_bwd_f32 dtypes")
    ld_d3 = 0
    if dqkv_split3 is not None:
        ld_d3 = _rowmajor(dqkv_split3, "dqkv_split3")
        _req(dqkv_split3.dtype == BF16 and dqkv_split3.shape[0] >= B * S and dqkv_split3.shape[1] >= 9 * heads * 64, "attn_bwd_f32: dqkv_split3")
    _req(qkv.shape[0] >= B * S and qkv.shape[1] >= 3 * heads * 64 and dqkv.shape[0] >= B * S and dqkv.shape[1] >= 3 * heads * 64
         and ctx.shape[0] >= B * S and ctx.shape[1] >= heads * 64 and dctx.shape[0] >= B * S and dctx.shape[1] >= heads * 64
         and lse.numel() >= B * heads * S, "attn_bwd_f32 shapes")
    if key_bias is not None:
        _req(key_bias.dtype == F32 and key_bias.is_contiguous() and key_bias.numel() >= B * S, "key_bias f32 [B,S]")
    dp, ds = (0.0, 0) if dropout is None else (float(dropout[0]), int(dropout[1]) & 0xFFFFFFFF)
    check(_l.load().bsclip_attn_bwd_f32(_p(qkv), _rowmajor(qkv, "qkv"), _p(dctx), _rowmajor(dctx, "dctx"), _p(ctx), _rowmajor(ctx, "ctx"),
                                        _p(lse), B, S, heads, _p(key_bias), float(scale), _p(dqkv), _rowmajor(dqkv, "dqkv"),
                                        _p(dqkv_split3), ld_d3, dp, ds, _stream()))


def im2col_patch16(image, cols):
    B = image.shape[0]
    _req(image.dtype == F32 and image.is_contiguous() and tuple(image.shape[1:]) == (3, 224, 224), "image f32 [B,3,224,224]")
    _req(cols.dtype == BF16 and cols.is_contiguous() and cols.shape[0] >= B * 196 and cols.shape[1] in (768, 2304),
         "cols bf16 [B*196, 768] (or [B*196, 2304]: split-bf16 rows [hi | lo | hi])")
    check(_l.load().bsclip_im2col_patch16(_p(image), B, _p(cols), cols.shape[1], int(cols.shape[1] == 2304), _stream()))


def mask_to_bias(mask, bias):
    _req(mask.dtype == torch.int64 and mask.is_contiguous() and bias.dtype == F32 and bias.is_contiguous()
         and bias.numel() >= mask.numel(), "mask_to_bias: int64 mask, f32 bias")
    check(_l.load().bsclip_mask_to_bias(_p(mask), mask.numel(), _p(bias), _stream()))


def vit_cls_rows(x, cls_token, pos_embed, B, S, H):
    _req(x.dtype in (F32, BF16) and x.is_contiguous() and x.numel() >= B * S * H, "x f32 / bf16 [B*S,H]")
    _req(cls_token.numel() == H and pos_embed.numel() >= H and cls_token.dtype == F32 and pos_embed.dtype == F32, "cls/pos")
    check(_l.load().bsclip_vit_cls_rows(_p(x), int(x.dtype == BF16), _p(cls_token), _p(pos_embed), B, S, H, _stream()))


def bert_embed(ids, type_ids, word, pos, typ, out):
    B, S = ids.shape
    H = word.shape[1]
    _req(ids.dtype == torch.int64 and ids.is_contiguous(), "ids int64 [B,S]")
    if type_ids is not None:
        _req(type_ids.dtype == torch.int64 and type_ids.is_contiguous() and type_ids.shape == ids.shape, "type_ids")
    _req(all(t.dtype == F32 and t.is_contiguous() for t in (word, pos, typ, out)), "bert_embed tables f32")
    _req(pos.shape[0] >= S and pos.shape[1] == H and typ.shape[0] >= 2 and typ.shape[1] == H, "bert_embed: pos/type shapes")
    _req(out.shape[0] >= B * S and out.shape[1] == H, "bert_embed: out f32 [B*S,H]")
    check(_l.load().bsclip_bert_embed(_p(ids), _p(type_ids), B, S, H, _p(word), word.shape[0], _p(pos), _p(typ), _p(out),
                                      _stream()))


def softmax_meanpool_fwd(logits, B, S, pooled, stats):
    C = logits.shape[1]
    _req(logits.dtype == F32 and logits.is_contiguous() and logits.shape[0] >= B * S, "logits f32 [B*S,C]")
    _req(pooled.dtype == F32 and pooled.is_contiguous() and pooled.shape[0] >= B and pooled.shape[1] == C, "pooled")
    _req(stats.dtype == F32 and stats.numel() >= 2 * B * S, "stats")
    check(_l.load().bsclip_softmax_meanpool_fwd(_p(logits), B, S, C, _p(pooled), _p(stats), _stream()))


def softmax_meanpool_bwd(logits, stats, d_pooled, B, S, dlogits):
    C = logits.shape[1]
    _req(logits.dtype == F32 and logits.is_contiguous() and d_pooled.dtype == F32 and d_pooled.is_contiguous(), "dtypes")
    _req(d_pooled.shape[0] >= B and d_pooled.shape[1] == C, "d_pooled f32 [B,C]")
    _req(dlogits.dtype == BF16 and dlogits.shape[0] >= B * S and dlogits.shape[1] >= C, "dlogits bf16 [B*S,C]")
    check(_l.load().bsclip_softmax_meanpool_bwd(_p(logits), _p(stats), _p(d_pooled), B, S, C, _p(dlogits),
                                                _rowmajor(dlogits, "dlogits"), _stream()))


def meanpool_tokens_fwd(x, B, S, out):
    H = x.shape[1]
    _req(x.dtype == F32 and x.is_contiguous() and x.shape[0] >= B * S, "x f32 [B*S,H]")
    _req(out.dtype == BF16 and out.shape[0] >= B and out.shape[1] >= H, "out bf16 [B,H]")
    check(_l.load().bsclip_meanpool_tokens_fwd(_p(x), B, S, H, _p(out), _rowmajor(out, "out"), _stream()))


def meanpool_tokens_bwd(d_pooled, B, S, dx):
    H = dx.shape[1]
    _req(d_pooled.dtype == F32 and d_pooled.shape[0] >= B and d_pooled.shape[1] >= H, "d_pooled f32 [B,>=H]")
    _req(dx.dtype == F32 and dx.is_contiguous() and dx.shape[0] >= B * S, "dx f32 [B*S,H]")
    check(_l.load().bsclip_meanpool_tokens_bwd(_p(d_pooled), _rowmajor(d_pooled, "d_pooled"), B, S, H, _p(dx), _stream()))


def dgelu_mul(g, z, M, N, out):
    _req(all(t.dtype == BF16 and t.shape[0] >= M and t.shape[1] >= N for t in (g, out)), "dgelu_mul: g/out bf16 [M,>=N]")
    _req(z.dtype == torch.uint8 and z.shape[0] >= M and z.shape[1] >= N, "dgelu_mul: z uint8 codes [M,>=N]")
    check(_l.load().bsclip_dgelu_mul(_p(g), _rowmajor(g, "g"), _p(z), _rowmajor(z, "z"), M, N, _p(out),
                                     _rowmajor(out, "out"), _stream()))


def l2norm_fwd(x, y, inv_norm):
    M, D = x.shape
    _req(all(t.dtype == F32 and t.is_contiguous() for t in (x, y, inv_norm)), "l2norm_fwd f32 contiguous")
    _req(y.shape == x.shape and inv_norm.numel() >= M, "l2norm_fwd shapes")
    check(_l.load().bsclip_l2norm_fwd(_p(x), M, D, _p(y), _p(inv_norm), _stream()))


def l2norm_bwd(y, inv_norm, dy, dx):
    M, D = y.shape
    _req(all(t.dtype == F32 and t.is_contiguous() for t in (y, inv_norm, dy, dx)), "l2norm_bwd f32 contiguous")
    _req(dy.shape == y.shape and dx.shape == y.shape and inv_norm.numel() >= M, "l2norm_bwd shapes")
    check(_l.load().bsclip_l2norm_bwd(_p(y), _p(inv_norm), _p(dy), M, D, _p(dx), _stream()))


def infonce_set_impl(impl):
    """0 = fused InfoNCE epilogues (default), 1 = f32 logits slabs (kept for timing / cross-checks)."""
    check(_l.load().bsclip_infonce_set_impl(int(impl)))


def infonce_workspace_floats(N, nmod):
    n = _l.load().bsclip_infonce_workspace_floats(N, nmod)
    if n < 0:
        raise ValueError("Too less element for calculating the contrastive loss." if nmod < 2 else "bad N/nmod")
    return n


def infonce_fwd_bwd(zs, labels, scale, loss_out, dzs=None, row0=0, n_local=None, workspace=None):
    """zs: list of 2-3 f32 [N, 768]; dzs: list of f32 [n_local, 768] (or None for loss only)."""
    nmod = len(zs)
    if nmod < 2:
        raise ValueError("Too less element for calculating the contrastive loss.")
    N, D = zs[0].shape
    n_local = N if n_local is None else n_local
    _req(all(z.dtype == F32 and z.is_contiguous() and tuple(z.shape) == (N, D) for z in zs), "zs f32 [N,D]")
    _req(labels.dtype == torch.int64 and labels.is_contiguous() and labels.numel() == N, "labels int64 [N]")
    _req(loss_out.dtype == F32 and loss_out.numel() >= 1, "loss_out f32 [1]")
    need = infonce_workspace_floats(N, nmod)
    _req(workspace is not None and workspace.dtype == F32 and workspace.numel() >= need and workspace.is_contiguous(),
         f"workspace must be f32 with >= {need} elements")
    zp = (ctypes.c_void_p * nmod)(*[z.data_ptr() for z in zs])
    dzp = None
    if dzs is not None:
        _req(len(dzs) == nmod and all(d.dtype == F32 and d.is_contiguous() and tuple(d.shape) == (n_local, D) for d in dzs),
             "dzs f32 [n_local, D]")
        dzp = (ctypes.c_void_p * nmod)(*[d.data_ptr() for d in dzs])
    check(_l.load().bsclip_infonce_fwd_bwd(zp, nmod, _p(labels), N, D, float(scale), row0, n_local, _p(loss_out), dzp,
                                           _p(workspace), _stream()))


def topk_ip(queries, keys, k):
    """Top-k inner products of L2-normalised rows: returns (scores f32 [Q,k], indices int64 [Q,k]) on the GPU."""
    Q, D = queries.shape
    K = keys.shape[0]
    _req(queries.dtype == F32 and keys.dtype == F32 and queries.is_contiguous() and keys.is_contiguous()
         and keys.shape[1] == D and queries.is_cuda and keys.is_cuda, "topk_ip: contiguous f32 GPU [Q,D], [K,D]")
    _req(1 <= k <= min(16, K) and D % 64 == 0, "topk_ip: 1 <= k <= 16, D % 64 == 0")
    ws = torch.empty(_l.load().bsclip_topk_ip_workspace_floats(Q, K, D), dtype=F32, device=queries.device)
    scores = torch.empty(Q, k, dtype=F32, device=queries.device)
    idx = torch.empty(Q, k, dtype=torch.int64, device=queries.device)
    check(_l.load().bsclip_topk_ip(_p(queries), Q, _p(keys), K, D, k, _p(scores), _p(idx), _p(ws), _stream()))
    return scores, idx


_LG_WS = {}


def _lora_grad_workspace(H, device):
    # one workspace per launching stream: the towers run concurrently on their own streams (SimpleCLIP.forward)
    key = (H, str(device), torch.cuda.current_stream().cuda_stream)
    ws = _LG_WS.get(key)
    if ws is None:
        ws = torch.empty(_l.load().bsclip_lora_grad_workspace_floats(H), dtype=F32, device=device)
        _LG_WS[key] = ws
    return ws


def lora_grad(dqkv, h_aug, M, H, lora_b, dt, dA, dBq, dBv):
    _req(dqkv.dtype == BF16 and h_aug.dtype == BF16, "lora_grad: bf16 inputs")
    _req(dqkv.shape[0] >= M and dqkv.shape[1] >= 3 * H and h_aug.shape[0] >= M and h_aug.shape[1] >= H + 8, "lora_grad shapes")
    _req(lora_b.dtype == F32 and lora_b.is_contiguous() and tuple(lora_b.shape) == (2, H, 4), "lora_b f32 [2,H,4]")
    _req(dt.dtype == F32 and dt.is_contiguous() and dt.numel() >= 8 * M, "dt f32 [M,8]")
    _req(dA.dtype == F32 and dA.is_contiguous() and tuple(dA.shape) == (8, H), "dA f32 [8,H]")
    _req(all(t.dtype == F32 and t.is_contiguous() and tuple(t.shape) == (H, 4) for t in (dBq, dBv)), "dB f32 [H,4]")
    check(_l.load().bsclip_lora_grad(_p(dqkv), _rowmajor(dqkv, "dqkv"), _p(h_aug), _rowmajor(h_aug, "h_aug"), M, H,
                                     _p(lora_b), _p(dt), _p(dA), _p(dBq), _p(dBv),
                                     _p(_lora_grad_workspace(H, dqkv.device)), _stream()))


def lora_grad_heads(h_aug, M, H, B, dt_partial, db_partial, dt, dA, dBq, dBv):
    """``lora_grad`` from the partial sums ``attn_bwd(..., lora=...)`` left: dt [M, 8], dBq / dBv and dA accumulate as ``lora_grad`` (dt_partial f32 [heads, 2, M, 4], db_partial f32 [B heads, 2, 4, 64])."""
    heads = H // 64
    _req(h_aug.dtype == BF16 and h_aug.shape[0] >= M and h_aug.shape[1] >= H and M % B == 0, "lora_grad_heads: h_aug bf16 [M, >= H]")
    _req(dt_partial.dtype == F32 and dt_partial.is_contiguous() and dt_partial.numel() >= heads * M * 8
         and db_partial.dtype == F32 and db_partial.is_contiguous() and db_partial.numel() >= B * heads * 512, "lora_grad_heads: partials")
    _req(dt.dtype == F32 and dt.is_contiguous() and dt.numel() >= 8 * M, "dt f32 [M,8]")
    _req(dA.dtype == F32 and dA.is_contiguous() and tuple(dA.shape) == (8, H), "dA f32 [8,H]")
    _req(all(t.dtype == F32 and t.is_contiguous() and tuple(t.shape) == (H, 4) for t in (dBq, dBv)), "dB f32 [H,4]")
    check(_l.load().bsclip_lora_grad_heads(_p(h_aug), _rowmajor(h_aug, "h_aug"), M, H, B, _p(dt_partial), _p(db_partial), _p(dt), _p(dA),
                                           _p(dBq), _p(dBv), _p(_lora_grad_workspace(H, h_aug.device)), _stream()))


def lora_grad_fp8(dqkv, y_fp8, t_aug, M, H, lora_b, dt, dA, dBq, dBv):
    _req(dqkv.dtype == BF16 and y_fp8.dtype == FP8 and t_aug.dtype == BF16, "lora_grad_fp8: dtypes")
    _req(dqkv.shape[0] >= M and dqkv.shape[1] >= 3 * H and y_fp8.shape[0] >= M and y_fp8.shape[1] >= H
         and t_aug.shape[0] >= M and t_aug.shape[1] >= 8, "lora_grad_fp8 shapes")
    _req(lora_b.dtype == F32 and lora_b.is_contiguous() and tuple(lora_b.shape) == (2, H, 4), "lora_b f32 [2,H,4]")
    _req(dt.dtype == F32 and dt.is_contiguous() and dt.numel() >= 8 * M, "dt f32 [M,8]")
    _req(dA.dtype == F32 and dA.is_contiguous() and tuple(dA.shape) == (8, H), "dA f32 [8,H]")
    _req(all(t.dtype == F32 and t.is_contiguous() and tuple(t.shape) == (H, 4) for t in (dBq, dBv)), "dB f32 [H,4]")
    check(_l.load().bsclip_lora_grad_fp8(_p(dqkv), _rowmajor(dqkv, "dqkv"), _p(y_fp8), _rowmajor(y_fp8, "y_fp8"), _p(t_aug),
                                         _rowmajor(t_aug, "t_aug"), M, H, _p(lora_b), _p(dt), _p(dA), _p(dBq), _p(dBv),
                                         _p(_lora_grad_workspace(H, dqkv.device)), _stream()))


def colsum(g, M, N, out):
    _req(g.dtype in (BF16, F32) and g.shape[0] >= M and g.shape[1] >= N, "colsum: g")
    _req(out.dtype == F32 and out.is_contiguous() and out.numel() >= N, "colsum: out f32 [N]")
    check(_l.load().bsclip_colsum(_p(g), _rowmajor(g, "g"), int(g.dtype == BF16), M, N, _p(out), _stream()))


def transpose_bf16(src, R, C, dst):
    _req(src.dtype == BF16 and dst.dtype == BF16, "transpose_bf16 dtypes")
    _req(src.shape[0] >= R and src.shape[1] >= C and dst.shape[0] >= C and dst.shape[1] >= R, "transpose_bf16 shapes")
    check(_l.load().bsclip_transpose_bf16(_p(src), _rowmajor(src, "src"), R, C, _p(dst), _rowmajor(dst, "dst"), _stream()))


_TC_WS = {}


def transpose_colsum_bf16(src, R, C, dst, colsum_out):
    """dst[C, R] = src[R, C]^T and colsum_out[c] += sum_r src[r, c] in one pass over src (weight-gradient operand + bias gradient)."""
    _req(src.dtype == BF16 and dst.dtype == BF16, "transpose_colsum_bf16 dtypes")
    _req(src.shape[0] >= R and src.shape[1] >= C and dst.shape[0] >= C and dst.shape[1] >= R, "transpose_colsum_bf16 shapes")
    _req(colsum_out.dtype == F32 and colsum_out.is_contiguous() and colsum_out.numel() >= C, "transpose_colsum_bf16: colsum f32 [C]")
    lib = _l.load()
    need = lib.bsclip_transpose_colsum_workspace_floats(R, C)
    key = (str(src.device), torch.cuda.current_stream().cuda_stream)
    ws = _TC_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _TC_WS[key] = torch.empty(need, dtype=F32, device=src.device)
    check(lib.bsclip_transpose_colsum_bf16(_p(src), _rowmajor(src, "src"), R, C, _p(dst), _rowmajor(dst, "dst"), _p(colsum_out),
                                           _p(ws), _stream()))


def cast_f32_bf16(src, dst):
    _req(src.dtype == F32 and dst.dtype == BF16 and src.is_contiguous() and dst.is_contiguous()
         and dst.numel() >= src.numel(), "cast_f32_bf16")
    check(_l.load().bsclip_cast_f32_bf16(_p(src), src.numel(), _p(dst), _stream()))


def waug_set_lora_layers(table, layers, ld_w, H):
    """One launch for all LoRA layers of an encoder: ``table`` int64 [layers, 3] of device addresses (W_aug, B_q, B_v)."""
    _req(table.dtype == torch.int64 and table.is_contiguous() and table.is_cuda and table.numel() >= 3 * layers, "waug_set_lora_layers: table")
    check(_l.load().bsclip_waug_set_lora_layers(_p(table), int(layers), int(ld_w), int(H), _stream()))


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    _req(all(t.dtype == F32 and t.is_contiguous() and t.numel() == p.numel() for t in (p, g, m, v)), "adamw: flat f32 buffers")
    check(_l.load().bsclip_adamw_step(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2),
                                      float(eps), float(weight_decay), int(step), float(grad_scale), _stream()))


def adamw_step_dev(p, g, m, v, hyper, beta1, beta2, eps, weight_decay, grad_scale=1.0):
    """AdamW with (lr, step) read from the device pair ``hyper`` = [lr as f32, step as uint32] when the kernel runs:
    graph-capturable (advance the step word with ``counter_add(hyper[1:2])``)."""
    _req(all(t.dtype == F32 and t.is_contiguous() and t.numel() == p.numel() for t in (p, g, m, v)), "adamw: flat f32 buffers")
    _req(hyper.element_size() == 4 and hyper.is_cuda and hyper.is_contiguous() and hyper.numel() >= 2,
         "adamw: hyper must be a device pair of 4-byte words (lr f32, step uint32)")
    check(_l.load().bsclip_adamw_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), float(beta1), float(beta2),
                                          float(eps), float(weight_decay), float(grad_scale), _stream()))


def set_dropout_step(counter):
    """Every dropout-carrying launch this thread makes from now on mixes the device word ``counter`` (uint32/int32 [1]) into
    its seed at run time (None: seeds are used as passed)."""
    if counter is not None:
        _req(counter.is_cuda and counter.numel() >= 1 and counter.element_size() == 4, "dropout step: 4-byte device word")
    check(_l.load().bsclip_set_dropout_step(_p(counter)))


def count_nonfinite(t, counter):
    """counter (int32 / uint32 device word) += the number of Inf / NaN elements of the contiguous f32 / bf16 tensor ``t``."""
    _req(t.dtype in (F32, BF16) and t.is_contiguous() and t.numel() > 0, "count_nonfinite: contiguous f32 / bf16 tensor")
    _req(counter.dtype == torch.int32 and counter.numel() >= 1 and counter.device == t.device, "count_nonfinite: int32 device counter")
    check(_l.load().bsclip_count_nonfinite(_p(t), t.numel(), int(t.dtype == BF16), _p(counter), _stream()))


def counter_add(counter, inc=1):
    _req(counter.is_cuda and counter.numel() >= 1 and counter.element_size() == 4, "counter: 4-byte device word")
    check(_l.load().bsclip_counter_add(_p(counter), int(inc) & 0xFFFFFFFF, _stream()))


def clock_probe(out32):
    """out32: zeroed int64 [32] on the GPU <- per XCD {shader-clock counter, 100 MHz real-time counter} when the current stream
    gets here (include/bsclip.h)."""
    _req(out32.is_cuda and out32.dtype == torch.int64 and out32.numel() >= 32 and out32.is_contiguous(), "clock_probe: int64 [32] on the GPU")
    check(_l.load().bsclip_clock_probe(_p(out32), _stream()))


def engine_clock_ghz(probe0, probe1):
    """Median over the XCDs of the shader clock between two clock_probe() samples, in GHz (None if no XCD gives a sane reading)."""
    a, b = probe0.cpu().view(16, 2), probe1.cpu().view(16, 2)
    ghz = []
    for x in range(16):
        dc, dt = int(b[x, 0] - a[x, 0]), int(b[x, 1] - a[x, 1])
        if a[x, 1] and b[x, 1] and dt > 0 and 0.5 < dc / (dt * 10.0) < 3.0:
            ghz.append(dc / (dt * 10.0))
    ghz.sort()
    return ghz[len(ghz) // 2] if ghz else None


def kmer_tokenize(blob, offsets, B, max_len, k, ids):
    _req(blob.dtype == torch.uint8 and blob.is_cuda and blob.is_contiguous(), "kmer_tokenize: uint8 GPU byte buffer")
    _req(offsets.dtype == torch.int64 and offsets.is_cuda and offsets.numel() == B + 1, "kmer_tokenize: offsets int64 [B+1]")
    _req(ids.dtype == torch.int64 and ids.is_cuda and ids.is_contiguous() and tuple(ids.shape) == (B, max_len // k + 1),
         "kmer_tokenize: ids int64 [B, max_len/k + 1]")
    check(_l.load().bsclip_kmer_tokenize(_p(blob), _p(offsets), B, max_len, k, _p(ids), _stream()))


def augment_images(src, records, B, mid_capacity, mid, out_size, out):
    _req(src.dtype == torch.uint8 and src.is_cuda and src.is_contiguous(), "augment_images: uint8 GPU source buffer")
    _req(records.dtype == torch.int32 and records.is_cuda and records.numel() == 16 * B, "augment_images: records int32 [B,16]")
    _req(mid.dtype == F32 and mid.is_cuda and mid.numel() >= 3 * B * mid_capacity, "augment_images: mid f32 [B,3,cap]")
    _req(out.dtype == F32 and out.is_contiguous() and tuple(out.shape) == (B, 3, out_size, out_size), "augment_images: out")
    check(_l.load().bsclip_augment_images(_p(src), _p(records), B, mid_capacity, _p(mid), out_size, _p(out), _stream()))


# ------------------------------------------------------------------------------------------- full fine-tuning (8f-4)
_PG_WS = {}


def ln_param_grad(x, stats, mode, d_gamma, d_beta, g_resid=None, g_gemm=None, dt=None, lora_a=None, M=None, in_dropout=None):
    """d_gamma / d_beta += LayerNorm parameter gradients, dy assembled as layernorm_bwd does."""
    ld_x = _rowmajor(x, "x")
    H = d_gamma.numel()
    M = x.shape[0] if M is None else M
    _req(x.dtype in (F32, BF16) and x.shape[1] >= H and M <= x.shape[0], "ln_param_grad: bad x")
    _req(stats.dtype == F32 and stats.numel() >= 2 * M, "ln_param_grad: stats")
    _req(all(t.dtype == F32 and t.is_contiguous() and t.numel() == H for t in (d_gamma, d_beta)), "ln_param_grad: d_gamma/d_beta")
    ld_g = ld_gr = 0
    if g_resid is not None:
        ld_gr = _rowmajor(g_resid, "g_resid")
        _req(g_resid.dtype == F32 and g_resid.shape[0] >= M and g_resid.shape[1] >= H, "g_resid f32 [M,>=H]")
    if g_gemm is not None:
        ld_g = _rowmajor(g_gemm, "g_gemm")
        _req(g_gemm.dtype == BF16 and g_gemm.shape[0] >= M and g_gemm.shape[1] >= H, "g_gemm bf16 [M,>=H]")
    if dt is not None:
        _req(dt.dtype == F32 and dt.is_contiguous() and dt.numel() >= 8 * M and lora_a is not None
             and tuple(lora_a.shape) == (8, H) and lora_a.dtype == F32 and lora_a.is_contiguous(), "dt f32 [M,8], lora_a f32 [8,H]")
    key = (H, str(x.device), torch.cuda.current_stream().cuda_stream)
    ws = _PG_WS.get(key)
    if ws is None:
        ws = _PG_WS[key] = torch.empty(_l.load().bsclip_ln_param_grad_workspace_floats(H), dtype=F32, device=x.device)
    dp, ds = (0.0, 0) if in_dropout is None else (float(in_dropout[0]), int(in_dropout[1]) & 0xFFFFFFFF)
    check(_l.load().bsclip_ln_param_grad(_p(x), ld_x, int(x.dtype == BF16), _p(stats), M, H, _p(g_resid), ld_gr, _p(g_gemm), ld_g,
                                         _p(dt), _p(lora_a) if dt is not None else None, int(mode), dp, ds, _p(d_gamma),
                                         _p(d_beta), _p(ws), _stream()))


def embed_grad(ids, type_ids, d_emb, d_word, d_pos, d_type, pad_id=0):
    B, S = ids.shape
    H = d_word.shape[1]
    _req(ids.dtype == torch.int64 and ids.is_contiguous(), "embed_grad: ids int64 [B,S]")
    _req(type_ids is None or (type_ids.dtype == torch.int64 and type_ids.is_contiguous() and type_ids.shape == ids.shape), "type_ids")
    _req(d_emb.dtype == F32 and d_emb.is_contiguous() and d_emb.shape[0] >= B * S and d_emb.shape[1] == H, "d_emb f32 [B*S,H]")
    _req(all(t.dtype == F32 and t.is_contiguous() and t.shape[1] == H for t in (d_word, d_pos, d_type)) and d_pos.shape[0] >= S
         and d_type.shape[0] >= 2, "embed_grad: gradient tables f32 [*,H]")
    key = ("embed", H, str(d_emb.device), torch.cuda.current_stream().cuda_stream)
    ws = _PG_WS.get(key)
    if ws is None:
        ws = _PG_WS[key] = torch.empty(_l.load().bsclip_embed_grad_workspace_floats(H), dtype=F32, device=d_emb.device)
    check(_l.load().bsclip_embed_grad(_p(ids), _p(type_ids), B, S, H, d_word.shape[0], int(pad_id), _p(d_emb), _p(d_word),
                                      _p(d_pos), _p(d_type), _p(ws), _stream()))


def gemm_splitk_f32(a, b, c, splits, partial, K=None):
    """c[M, N] (f32) += a[M, K] . b[N, K]^T, reduction cut into ``splits`` K ranges (weight gradients of full fine-tuning)."""
    M, N = c.shape
    K = a.shape[1] if K is None else K
    lda, ldb, ldc = _rowmajor(a, "a"), _rowmajor(b, "b"), _rowmajor(c, "c")
    _req(a.dtype == BF16 and b.dtype == BF16 and c.dtype == F32 and partial.dtype == F32 and partial.is_contiguous(), "gemm_splitk_f32 dtypes")
    _req(a.shape[0] >= M and b.shape[0] >= N and a.shape[1] >= K and b.shape[1] >= K, "gemm_splitk_f32: operand shapes")
    _req(N % 256 == 0 and splits >= 1 and K % (64 * splits) == 0, "gemm_splitk_f32: N % 256 == 0, K % (64 * splits) == 0")
    _req(partial.numel() >= splits * M * N, "gemm_splitk_f32: partial needs splits * M * N floats")
    check(_l.load().bsclip_gemm_splitk_f32(_p(a), lda, _p(b), ldb, _p(c), ldc, M, N, K, int(splits), _p(partial), _stream()))


def gather_cast_rows(src, rows_out, period_in, period_out, offset, dst):
    H = src.shape[1]
    _req(src.dtype == F32 and dst.dtype == BF16 and dst.shape[0] >= rows_out and dst.shape[1] >= H, "gather_cast_rows dtypes/shapes")
    n_in = (rows_out + period_out - 1) // period_out * period_in
    _req(src.shape[0] >= n_in - (period_in - offset - period_out), "gather_cast_rows: src too small")
    check(_l.load().bsclip_gather_cast_rows(_p(src), _rowmajor(src, "src"), rows_out, period_in, period_out, offset, H, _p(dst),
                                            _rowmajor(dst, "dst"), _stream()))
