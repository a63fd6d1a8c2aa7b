"""Tests of the round-4 experiment kernels (key-owner-sweep attention backward, bsclip_attn_fwd2 / bsclip_attn_bwd2), which left the
product library with ABI 9 and live in libbsclip_hip_diag.so only.  Not part of the driver's suite (tests/): run by hand with
    make -C bioscan-clip_amd/csrc diag && python -m pytest tools/experiments -q
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bioscan-clip_amd"), os.path.join(ROOT, "tests"), os.path.dirname(os.path.abspath(__file__))]
from helpers import rel_err  # noqa: E402

PD = 0.1


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bioscanclip.hip import lib, ops as o
    if not os.path.exists(lib.DIAG_LIB_PATH):
        pytest.skip("libbsclip_hip_diag.so is not built (make -C bioscan-clip_amd/csrc diag)")
    return o


import attn_sweep_ops as xo  # noqa: E402


def dev(t):
    return t.to("cuda")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _attn_ref(qkv, B, S, heads, scale, bias):
    H = heads * 64
    q, k, v = [t.reshape(B, S, heads, 64).transpose(1, 2) for t in qkv.split(H, dim=-1)]
    s = (q @ k.transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias[:, None, None, :]
    p = torch.softmax(s, -1)
    return (p @ v).transpose(1, 2).reshape(B * S, H), torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,S,heads,masked", [(3, 197, 12, False), (2, 133, 12, False), (5, 20, 8, True),
                                              (2, 64, 2, False), (1, 33, 1, False), (2, 7, 3, True),
                                              (2, 224, 2, True), (3, 1, 2, False), (1, 193, 1, True),
                                              # more (batch, head) items than CUs: the persistent kernel walks 2 items per workgroup
                                              (30, 197, 12, False), (26, 133, 12, False), (44, 197, 12, False)])
def test_attention_sweep_fwd_bwd(ops, B, S, heads, masked):
    """bsclip_attn_fwd2 / bsclip_attn_bwd2 (key-owner sweep, delta from the forward's 16-bit output): same bars as the two-phase
    kernels against the f32 torch reference, the same ctx bit for bit, and dqkv within bf16 rounding of the two-phase result."""
    H = heads * 64
    qkv = dev(rnd(B * S, 3 * H + 64, seed=1).bfloat16())[:, :3 * H]
    bias = None
    if masked:
        lens = torch.randint(1, S + 1, (B,), generator=torch.Generator().manual_seed(3))
        m = (torch.arange(S)[None] < lens[:, None]).float()
        bias = dev((1.0 - m) * torch.finfo(torch.float32).min)
    scale = 0.125
    ctx, ctx_lo = (torch.full((B * S, H), float("nan"), device="cuda", dtype=torch.bfloat16) for _ in range(2))
    stats = torch.full((B, heads, S, 4), float("nan"), device="cuda")
    xo.attn_fwd2(qkv, B, S, heads, scale, ctx, ctx_lo, stats, key_bias=bias)
    ctx1 = torch.empty_like(ctx)
    lse = torch.empty(B, heads, S, device="cuda")
    ops.attn_fwd(qkv, B, S, heads, scale, ctx1, lse, key_bias=bias)
    assert torch.equal(ctx, ctx1)
    qf = qkv.float().reshape(B, S, 3 * H).requires_grad_(True)
    ref, ref_lse = _attn_ref(qf, B, S, heads, scale, bias)
    assert rel_err(ctx.float(), ref) < 6e-3
    assert rel_err(ctx.float() + ctx_lo.float(), ref) <= rel_err(ctx.float(), ref)     # the residual is a residual (S = 1: both 0)
    # stats: lse = (log2(1 / inv) - nm2) ln 2; rZ within bf16 rounding of 1
    lse2 = (torch.log2(1.0 / stats[..., 1]) - stats[..., 0]) * 0.6931471805599453
    assert rel_err(lse2, ref_lse) < 1e-5
    assert (stats[..., 2] - 1).abs().max() < 4e-3

    dctx = dev(rnd(B * S, H, seed=2).bfloat16())
    (gq,) = torch.autograd.grad(ref, qf, dctx.float())
    dqkv = torch.full((B * S, 3 * H), float("nan"), device="cuda", dtype=torch.bfloat16)
    xo.attn_bwd2(qkv, dctx, ctx, ctx_lo, stats, B, S, heads, scale, dqkv, key_bias=bias)
    old = torch.empty_like(dqkv)
    ops.attn_bwd(qkv, dctx, lse, B, S, heads, scale, old, key_bias=bias)
    gq = gq.reshape(B * S, 3 * H)
    # S = 1: dQ = dK = 0 exactly in the reference (one key: dS = P (dP - delta) = 0); here delta comes from a different f32
    # summation than dP, so dS is f32 rounding noise instead of an exact zero: errors are measured against the whole gradient
    floor = 1e-5 * gq.norm()
    err = lambda a, ref: ((a - ref).norm() / (ref.norm() + floor)).item()
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        e_new, e_old = err(dqkv[:, sl].float(), gq[:, sl]), err(old[:, sl].float(), gq[:, sl])
        assert e_new < 1e-2, (name, e_new)
        assert S == 1 or e_new < 1.15 * e_old + 1e-4, (name, e_new, e_old)   # S = 1: the two-phase kernel's dS is an exact 0


def test_attention_sweep_keeps_the_softmax_backward_cancellation(ops):
    """Values nearly equal across keys: dP_k is almost constant over k and dS = P (dP - delta) is a small difference of large
    numbers.  delta from the bf16-rounded output alone would lose it (gradient error ~ 10-50 %); from O to 16 bits and the rounded
    operand's own normaliser Z' it matches the two-phase kernel, which recomputes delta from the f32 tiles."""
    B, S, heads = 2, 197, 4
    H = heads * 64
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B * S, H, generator=g)
    k = torch.randn(B * S, H, generator=g) * 3.0                       # peaked attention rows
    v = torch.randn(1, H, generator=g) + 0.01 * torch.randn(B * S, H, generator=g)
    qkv = dev(torch.cat([q, k, v], dim=1).bfloat16())
    ctx, ctx_lo = (torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    stats = torch.empty(B, heads, S, 4, device="cuda")
    xo.attn_fwd2(qkv, B, S, heads, 0.125, ctx, ctx_lo, stats)
    dctx = dev(rnd(B * S, H, seed=2).bfloat16())
    dqkv = torch.empty(B * S, 3 * H, device="cuda", dtype=torch.bfloat16)
    xo.attn_bwd2(qkv, dctx, ctx, ctx_lo, stats, B, S, heads, 0.125, dqkv)
    qf = qkv.double().reshape(B, S, 3 * H).requires_grad_(True)
    ref, _ = _attn_ref(qf, B, S, heads, 0.125, None)
    (gq,) = torch.autograd.grad(ref, qf, dctx.double())
    gq = gq.reshape(B * S, 3 * H).float()
    assert rel_err(dqkv[:, :H].float(), gq[:, :H]) < 1.5e-2 and rel_err(dqkv[:, H:2 * H].float(), gq[:, H:2 * H]) < 1.5e-2
    # ... and the experiment that shows the bar has teeth: the same call with the residual withheld
    xo.attn_bwd2(qkv, dctx, ctx, torch.zeros_like(ctx_lo), stats, B, S, heads, 0.125, dqkv)
    assert rel_err(dqkv[:, :H].float(), gq[:, :H]) > 5e-2


@pytest.mark.parametrize("B,S,heads,masked", [(2, 64, 2, False), (3, 20, 8, True), (2, 133, 3, False), (2, 197, 3, False),
                                              (25, 133, 12, False)])     # 300 items: the persistent kernel, 2 per workgroup
def test_attention_sweep_dropout_matches_two_phase(ops, B, S, heads, masked):
    """bsclip_attn_fwd2 / bwd2 with attention-probs dropout: the same (seed, element) masks as the two-phase kernels -- the
    forward output is identical bit for bit -- and a gradient that agrees with theirs (which the test above holds to torch
    autograd on the extracted mask) within bf16 rounding, dropped keys included (dS = -P delta there)."""
    H = heads * 64
    qkv = rnd(B * S, 3 * H, seed=1).bfloat16().cuda()
    bias = None
    if masked:
        lens = torch.randint(2, S + 1, (B,), generator=torch.Generator().manual_seed(3))
        bias = ((1.0 - (torch.arange(S)[None] < lens[:, None]).float()) * torch.finfo(torch.float32).min).cuda()
    seed = 4242
    ctx, ctx_lo, ctx1 = (torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    stats = torch.empty(B, heads, S, 4, device="cuda")
    lse = torch.empty(B, heads, S, device="cuda")
    ops.attn_fwd(qkv, B, S, heads, 0.125, ctx1, lse, key_bias=bias, dropout=(PD, seed))
    xo.attn_fwd2(qkv, B, S, heads, 0.125, ctx, ctx_lo, stats, key_bias=bias, dropout=(PD, seed))
    assert torch.equal(ctx, ctx1)
    dctx = rnd(B * S, H, seed=2).bfloat16().cuda()
    new, old = (torch.empty(B * S, 3 * H, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    ops.attn_bwd(qkv, dctx, lse, B, S, heads, 0.125, old, key_bias=bias, dropout=(PD, seed))
    xo.attn_bwd2(qkv, dctx, ctx, ctx_lo, stats, B, S, heads, 0.125, new, key_bias=bias, dropout=(PD, seed))
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        assert rel_err(new[:, sl].float(), old[:, sl].float()) < 8e-3, (name, rel_err(new[:, sl].float(), old[:, sl].float()))
