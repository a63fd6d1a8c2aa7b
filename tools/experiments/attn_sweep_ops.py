"""Host wrappers of the round-4 experiment kernels (bsclip_attn_fwd2 / bsclip_attn_bwd2: key-owner-sweep attention backward and its
persistent form).  They are NOT product kernels (ABI 9): they live in libbsclip_hip_diag.so only (`make -C bioscan-clip_amd/csrc diag`)
and are reached from tools/ and tools/experiments/test_attn_sweep_gpu.py, never from the engines."""
from bioscanclip.hip import lib as _l
from bioscanclip.hip.ops import BF16, F32, _p, _req, _rowmajor, _stream, check


def attn_fwd2(qkv, B, S, heads, scale, ctx, ctx_lo, stats, key_bias=None, dropout=None):
    """Forward of the key-owner-sweep backward (``attn_bwd2``): ctx = bf16(O), ctx_lo = bf16(O - ctx), stats f32 [B, heads, S, 4]."""
    ld_qkv, ld_ctx = _rowmajor(qkv, "qkv"), _rowmajor(ctx, "ctx")
    _req(qkv.dtype == BF16 and ctx.dtype == BF16 and ctx_lo.dtype == BF16 and stats.dtype == F32, "attn_fwd2 dtypes")
    _req(_rowmajor(ctx_lo, "ctx_lo") == ld_ctx and ctx_lo.shape == ctx.shape, "attn_fwd2: ctx_lo must have ctx's layout")
    _req(qkv.shape[0] >= B * S and qkv.shape[1] >= 3 * heads * 64, "attn_fwd2: qkv too small")
    _req(ctx.shape[0] >= B * S and ctx.shape[1] >= heads * 64 and stats.numel() >= B * heads * S * 4 and stats.is_contiguous(),
         "attn_fwd2: outputs too small")
    if key_bias is not None:
        _req(key_bias.dtype == F32 and key_bias.is_contiguous() and key_bias.numel() >= B * S, "key_bias f32 [B,S]")
    dp, ds = (0.0, 0) if dropout is None else (float(dropout[0]), int(dropout[1]) & 0xFFFFFFFF)
    check(_l.load_diag().bsclip_attn_fwd2(_p(qkv), ld_qkv, B, S, heads, _p(key_bias), float(scale), _p(ctx), _p(ctx_lo), ld_ctx,
                                     _p(stats), dp, ds, _stream()))


def attn_bwd2(qkv, dctx, ctx, ctx_lo, stats, B, S, heads, scale, dqkv, key_bias=None, dropout=None):
    ld_qkv, ld_dctx, ld_ctx, ld_d = _rowmajor(qkv, "qkv"), _rowmajor(dctx, "dctx"), _rowmajor(ctx, "ctx"), _rowmajor(dqkv, "dqkv")
    _req(all(t.dtype == BF16 for t in (qkv, dctx, ctx, ctx_lo, dqkv)) and stats.dtype == F32, "attn_bwd2 dtypes")
    _req(_rowmajor(ctx_lo, "ctx_lo") == ld_ctx and ctx_lo.shape == ctx.shape, "attn_bwd2: ctx_lo must have ctx's layout")
    _req(min(qkv.shape[0], dctx.shape[0], dqkv.shape[0], ctx.shape[0]) >= B * S, "attn_bwd2: rows")
    _req(qkv.shape[1] >= 3 * heads * 64 and dqkv.shape[1] >= 3 * heads * 64 and dctx.shape[1] >= heads * 64
         and ctx.shape[1] >= heads * 64 and stats.numel() >= B * heads * S * 4 and stats.is_contiguous(), "attn_bwd2: cols")
    if key_bias is not None:
        _req(key_bias.dtype == F32 and key_bias.is_contiguous() and key_bias.numel() >= B * S, "key_bias f32 [B,S]")
    dp, ds = (0.0, 0) if dropout is None else (float(dropout[0]), int(dropout[1]) & 0xFFFFFFFF)
    check(_l.load_diag().bsclip_attn_bwd2(_p(qkv), ld_qkv, _p(dctx), ld_dctx, _p(ctx), _p(ctx_lo), ld_ctx, _p(stats), B, S, heads,
                                     _p(key_bias), float(scale), _p(dqkv), ld_d, dp, ds, _stream()))


# ---- "exact" forward mode (BSCLIP_PARITY=2): split-bf16 operands for every trunk GEMM, f32 attention (csrc/exact.hip) ----
