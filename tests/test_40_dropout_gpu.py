"""Dropout (HF BERT hidden 0.1 / attention-probs 0.1, active in train mode).  Bit-parity with torch's
Philox stream is impossible by construction, so correctness is established structurally:
  * every site keeps ~ (1-p) of its elements and scales survivors by 1/(1-p) (statistics);
  * the mask is a pure function of (seed, element index): the backward kernels regenerate exactly the mask the forward
    used (pattern equality / comparison with torch autograd run on the mask extracted from the forward)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import rel_err  # noqa: E402

P = 0.1


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bioscanclip.hip import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).cuda()


def test_gemm_resid_dropout_and_layernorm_bwd_share_the_mask(ops):
    from bioscanclip.hip.lib import EPI_RESID_F32
    M, N, K = 3000, 768, 768
    a, w = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, scale=0.05).bfloat16()
    bias, resid = rnd(N, seed=3), rnd(M, N, seed=4)
    ref = a.float() @ w.float().t() + bias
    outs = {}
    for tile in (1, 4, 5):
        ops.set_gemm_tile(tile)
        out = torch.empty(M, N, device="cuda")
        ops.gemm(a, w, out, EPI_RESID_F32, bias=bias, resid=resid, dropout=(P, 1234))
        outs[tile] = out - resid
    ops.set_gemm_tile(0)
    assert torch.equal(outs[1] == 0, outs[4] == 0) and torch.equal(outs[5] == 0, outs[4] == 0), "mask must not depend on the kernel/tile layout"
    from bioscanclip.hip.lib import EPI_RESID_BF16
    outb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, w, outb, EPI_RESID_BF16, bias=bias, resid=torch.zeros(M, N, device="cuda", dtype=torch.bfloat16), dropout=(P, 1234))
    assert torch.equal(outb == 0, outs[4] == 0), "the bf16-stream epilogue draws the same mask"
    d = outs[4]
    kept = d != 0
    assert abs(kept.float().mean().item() - (1 - P)) < 5e-3
    assert rel_err(d[kept], (ref / (1 - P))[kept]) < 1e-5
    out2 = torch.empty(M, N, device="cuda")
    ops.gemm(a, w, out2, EPI_RESID_F32, bias=bias, resid=resid, dropout=(P, 1234))
    assert torch.equal(out2 - resid, d), "same seed -> same mask"
    ops.gemm(a, w, out2, EPI_RESID_F32, bias=bias, resid=resid, dropout=(P, 99))
    assert ((out2 - resid == 0) != (d == 0)).float().mean() > 0.1, "different seed -> different mask"
    # backward: dx_bf16 of the LayerNorm that consumes this GEMM's output carries the same mask; dx_f32 does not
    x = rnd(M, N, seed=5)
    g, b = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    stats = torch.empty(M, 2, device="cuda")
    ops.layernorm_fwd(x, g, b, 1e-12, y_f32=torch.empty(M, N, device="cuda"), stats=stats)
    gres = rnd(M, N, seed=6)
    dx32, dx16 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.layernorm_bwd(x, stats, g, 1, g_resid=gres, dx_f32=dx32, dx_bf16=dx16, dropout=(P, 1234))
    assert torch.equal(dx16 == 0, ~kept)
    assert (dx32 != 0).all()
    assert rel_err(dx16.float()[kept], (dx32 / (1 - P))[kept]) < 4e-3


def test_masks_of_successive_steps_are_independent(ops):
    """with the step added linearly to the seed, step s+1's mask was step s's mask slid by one element pair.  The
    step word now goes through the mixer: masks of steps s and s+1 (and of two engines' seeds) agree at every small shift only
    as often as independent masks do (p^2 + (1-p)^2 = 0.82 at p = 0.1; a shifted copy agrees everywhere)."""
    from bioscanclip.hip.lib import EPI_RESID_F32
    M, N, K = 512, 768, 64
    a, w = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16), torch.zeros(N, K, device="cuda", dtype=torch.bfloat16)
    bias, resid = torch.ones(N, device="cuda"), torch.zeros(M, N, device="cuda")
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    masks = []
    try:
        ops.set_dropout_step(word)
        for s in range(4):
            ops.counter_add(word, 1)
            out = torch.empty(M, N, device="cuda")
            ops.gemm(a, w, out, EPI_RESID_F32, bias=bias, resid=resid, dropout=(P, 4321))
            masks.append((out != 0).flatten())
    finally:
        ops.set_dropout_step(None)
    want = P * P + (1 - P) * (1 - P)
    for s in range(3):
        m0, m1 = masks[s], masks[s + 1]
        assert abs(m1.float().mean().item() - (1 - P)) < 5e-3
        for shift in range(-8, 9):
            x, y = (m0[shift:], m1[:m1.numel() - shift]) if shift >= 0 else (m0[:shift], m1[-shift:])
            agree = (x == y).float().mean().item()
            assert abs(agree - want) < 1e-2, (s, shift, agree)


def test_layernorm_fwd_dropout(ops):
    M, H = 2000, 768
    x = rnd(M, H, seed=1)
    g, b = 1 + 0.1 * rnd(H, seed=2), 0.1 * rnd(H, seed=3) + 0.5
    A = rnd(8, H, seed=4, scale=0.05)
    y16 = torch.empty(M, H + 64, device="cuda", dtype=torch.bfloat16)
    y32 = torch.empty(M, H, device="cuda")
    ops.layernorm_fwd(x, g, b, 1e-12, y_bf16=y16, y_f32=y32, lora_a=A, dropout=(P, 7))
    ref = torch.nn.functional.layer_norm(x, (H,), g, b, 1e-12)
    kept = y32 != 0
    assert abs(kept.float().mean().item() - (1 - P)) < 5e-3
    assert rel_err(y32[kept], (ref / (1 - P))[kept]) < 1e-5
    assert torch.equal(y16[:, :H] == 0, ~kept)
    assert rel_err(y16[:, H:H + 8].float(), y32 @ A.t()) < 4e-3, "LoRA projection must see the dropped activations"


@pytest.mark.parametrize("B,S,heads,masked", [(2, 64, 2, False), (3, 20, 8, True), (2, 197, 3, False)])
def test_attention_dropout_fwd_bwd(ops, B, S, heads, masked):
    H = heads * 64
    scale = 0.125
    qkv = rnd(B * S, 3 * H, seed=1).bfloat16()
    bias = None
    if masked:
        lens = torch.randint(2, S + 1, (B,), generator=torch.Generator().manual_seed(3))
        bias = ((1.0 - (torch.arange(S)[None] < lens[:, None]).float()) * torch.finfo(torch.float32).min).cuda()
    seed = 4242
    # 1) extract the mask: with V = [I; 0] blocks the context rows spell out the dropped probabilities
    n_chunk = (S + 63) // 64
    mask = torch.zeros(B, heads, S, S, device="cuda")
    ptil = torch.zeros(B, heads, S, S, device="cuda")
    lse = torch.empty(B, heads, S, device="cuda")
    for c in range(n_chunk):
        probe = qkv.clone().float().reshape(B, S, 3, heads, 64)
        v = torch.zeros(B, S, heads, 64, device="cuda")
        ks = torch.arange(c * 64, min(S, c * 64 + 64), device="cuda")
        v[:, ks, :, ks - c * 64] = 1.0
        probe[:, :, 2] = v
        ctx = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16)
        ops.attn_fwd(probe.reshape(B * S, 3 * H).bfloat16(), B, S, heads, scale, ctx, lse, key_bias=bias, dropout=(P, seed))
        got = ctx.float().reshape(B, S, heads, 64).permute(0, 2, 1, 3)  # [B, heads, q, key-in-chunk]
        ptil[:, :, :, ks] = got[..., : len(ks)]
    q, k, v = [t.reshape(B, S, heads, 64).transpose(1, 2) for t in qkv.float().split(H, dim=-1)]
    s = (q @ k.transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias[:, None, None, :]
    p = torch.softmax(s, -1)
    visible = p > 1e-3                      # bf16 output cannot resolve tiny probabilities
    mask = (ptil != 0)
    frac = mask[visible].float().mean().item()
    assert abs(frac - (1 - P)) < 0.02, frac
    assert rel_err(ptil[visible & mask], (p / (1 - P))[visible & mask]) < 6e-3
    # 2) forward + backward with the real V against torch autograd driven by the extracted mask
    keepf = torch.where(visible, mask.float(), torch.ones_like(p)) / (1 - P)   # invisible entries: contribution ~0
    qf = qkv.float().reshape(B, S, 3 * H).requires_grad_(True)
    q2, k2, v2 = [t.reshape(B, S, heads, 64).transpose(1, 2) for t in qf.split(H, dim=-1)]
    s2 = (q2 @ k2.transpose(-1, -2)) * scale
    if bias is not None:
        s2 = s2 + bias[:, None, None, :]
    ref = ((torch.softmax(s2, -1) * keepf) @ v2).transpose(1, 2).reshape(B * S, H)
    ctx = torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16)
    ops.attn_fwd(qkv, B, S, heads, scale, ctx, lse, key_bias=bias, dropout=(P, seed))
    assert rel_err(ctx.float(), ref) < 2e-2
    dctx = rnd(B * S, H, seed=2).bfloat16()
    (gq,) = torch.autograd.grad(ref, qf, dctx.float())
    dqkv = torch.empty(B * S, 3 * H, device="cuda", dtype=torch.bfloat16)
    ops.attn_bwd(qkv, dctx, lse, B, S, heads, scale, dqkv, key_bias=bias, dropout=(P, seed))
    gq = gq.reshape(B * S, 3 * H)
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        assert rel_err(dqkv[:, sl].float(), gq[:, sl]) < 3e-2, name


@pytest.mark.parametrize("B,S,heads,masked", [(2, 64, 2, False), (3, 20, 8, True), (2, 133, 12, False), (2, 197, 3, False),
                                              (1, 224, 2, True), (3, 1, 2, False), (30, 133, 12, False)])
def test_attention_keep_bits_reproduce_the_hashed_masks(ops, B, S, heads, masked):
    """Round 5: the forward leaves its dropout decisions as bit words (64 B per query row) and the backward reads them instead of
    re-hashing every element.  Same decisions, same arithmetic: the forward output does not change, the gradients are bit for bit
    those of the re-hashing backward, and the words themselves spell the mask the test above extracts through V = I."""
    H = heads * 64
    qkv = rnd(B * S, 3 * H, seed=1).bfloat16()
    dctx = rnd(B * S, H, seed=2).bfloat16()
    bias = None
    if masked:
        lens = torch.randint(2, S + 1, (B,), generator=torch.Generator().manual_seed(3))
        bias = ((1.0 - (torch.arange(S)[None] < lens[:, None]).float()) * torch.finfo(torch.float32).min).cuda()
    seed = 4242
    ctx0, ctx1 = (torch.empty(B * S, H, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    lse0, lse1 = (torch.empty(B, heads, S, device="cuda") for _ in range(2))
    bits = torch.full((B * heads * S * ops.KEEP_WORDS,), -1, device="cuda", dtype=torch.int32)
    ops.attn_fwd(qkv, B, S, heads, 0.125, ctx0, lse0, key_bias=bias, dropout=(P, seed))
    ops.attn_fwd(qkv, B, S, heads, 0.125, ctx1, lse1, key_bias=bias, dropout=(P, seed), keep_bits=bits)
    assert torch.equal(ctx0, ctx1) and torch.equal(lse0, lse1)
    g0 = torch.full((B * S, 3 * H), float("nan"), device="cuda", dtype=torch.bfloat16)
    g1 = torch.full_like(g0, float("nan"))
    ops.attn_bwd(qkv, dctx, lse0, B, S, heads, 0.125, g0, key_bias=bias, dropout=(P, seed))
    ops.attn_bwd(qkv, dctx, lse0, B, S, heads, 0.125, g1, key_bias=bias, dropout=(P, seed), keep_bits=bits)
    assert torch.equal(g0, g1)
    # the words: bit 8 g + i of word kt of lane half h = key 32 kt + 8 g + 4 h + i; about 1 - p of the valid keys are kept
    w = bits.view(B * heads, S, 2, ops.KEEP_WORDS // 2).cpu().to(torch.int64) & 0xFFFFFFFF
    nb = (S + 31) // 32
    keep = torch.zeros(B * heads, S, nb * 32, dtype=torch.bool)
    for h in range(2):
        for kt in range(nb):
            for g in range(4):
                for i in range(4):
                    keep[:, :, 32 * kt + 8 * g + 4 * h + i] = ((w[:, :, h, kt] >> (8 * g + i)) & 1).bool()
    frac = keep[:, :, :S].float().mean().item()
    assert abs(frac - (1 - P)) < max(0.01, 4.0 * (P * (1 - P) / (B * heads * S * S)) ** 0.5), frac   # four sigma of the sample mean
    # ... and a different seed gives different words, the same seed the same words
    bits2, bits3 = torch.zeros_like(bits), torch.zeros_like(bits)
    ops.attn_fwd(qkv, B, S, heads, 0.125, ctx1, lse1, key_bias=bias, dropout=(P, seed), keep_bits=bits2)
    ops.attn_fwd(qkv, B, S, heads, 0.125, ctx1, lse1, key_bias=bias, dropout=(P, seed + 1), keep_bits=bits3)
    valid = torch.ones(B * heads, S, 2, ops.KEEP_WORDS // 2, dtype=torch.bool)
    valid[:, :, :, nb:] = False
    v = valid.view(-1).cuda()
    assert torch.equal(bits2[v], bits[v]) and (S == 1 or not torch.equal(bits3[v], bits[v]))


def test_engine_train_vs_eval_mode():
    """train mode: stochastic (a new mask per forward) with an unbiased mean; eval mode: deterministic and identical
    to the p = 0 configuration."""
    from oracle import synth
    from bioscanclip.model import arch
    from bioscanclip.model.language_encoder import LoRA_bert
    def build(**kw):
        m = LoRA_bert(arch.BertModelParams(arch.bert_small_config(**kw)), r=4, num_classes=768)
        sd = synth.synth_state_dict({k: v for k, v in synth.shapes_of(m).items()}, 12)
        m.load_state_dict(sd)
        return m.cuda()
    _, _, text, _ = synth.synth_batch(16, seed=22, with_text=True)
    text = {k: v.cuda() for k, v in text.items()}
    m_drop, m_nodrop = build(), build(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m_nodrop.train()
    ref = m_nodrop(text).detach()
    m_drop.eval()
    with torch.no_grad():
        assert torch.equal(m_drop(text), m_drop(text))
        assert rel_err(m_drop(text), ref) < 1e-6
    m_drop.train()
    y1 = m_drop(text)
    y2 = m_drop(text)
    assert rel_err(y1, y2) > 1e-3, "two train-mode forwards must use different masks"
    assert rel_err(y1, ref) < 0.5
    (y2 * torch.randn_like(y2)).sum().backward()
    grads = [p.grad for p in m_drop.parameters() if p.requires_grad]
    assert all(torch.isfinite(g).all() and g.abs().sum() > 0 for g in grads)


@pytest.mark.parametrize("full_ft,exact", [(False, False), (True, False), (False, True)])
def test_directional_derivative_with_the_masks_held_fixed(full_ft, exact, monkeypatch):
    """End-to-end check of the backward pass WITH dropout: the masks are a function of (site seed, element, step word), so with
    the engine's step word reset before every forward the encoder is a deterministic function of its parameters, and the
    gradient the kernels produce must agree with a central finite difference of  L = sum(y * cot)  along a random direction over
    all trainable tensors (the gradient direction, per tensor at the parameter's scale) -- LoRA pairs + decoder, or every parameter under full fine-tuning (weight, bias, LayerNorm,
    embedding gradients are then taken under the forward's masks, including the embedding LayerNorm's).  Three step sizes; the
    smallest must agree to 5 % (measured: 1.2 % in the LoRA regime, 1e-4 under full fine-tuning).  ``exact``: the same under
    BSCLIP_PARITY=2 -- the f32 attention forward / backward, the f32 LayerNorm operands and the split-operand GEMMs draw and
    re-apply the masks of the same three dropout sites."""
    from oracle import synth
    if exact:
        from bioscanclip.hip import engine
        monkeypatch.setattr(engine, "RESID_STREAM_BF16", False)
        monkeypatch.setattr(engine, "GRAD_STREAM_BF16", False)
        monkeypatch.setattr(engine, "EXACT_FORWARD", True)
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    m = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=2)), r=4, num_classes=768,
                          lora_layer=[] if full_ft else None)
    sd = synth.synth_state_dict({"dna_encoder." + k: v for k, v in synth.shapes_of(m).items()}, 11)
    m.load_state_dict({k[len("dna_encoder."):]: v for k, v in sd.items()})
    if full_ft:
        for p in m.parameters():
            p.requires_grad = True
        m.hip_full_ft = True
    torch.manual_seed(11)                       # the dropout site seeds derive from torch.initial_seed()
    m.cuda().train()
    x = synth.synth_batch(8, seed=21)[1].cuda()
    cot = synth.synth_tensor("fd.cot", (8, 768), seed=5).cuda()
    y = m(x)                                    # builds the engine

    def L(grad):
        m._engine._step_word.zero_()            # every forward below draws the masks of step 1
        if grad:
            for p in m.parameters():
                p.grad = None
            y = m(x)
            (y * cot).sum().backward()
            return (y.detach() * cot).sum().item()
        with torch.no_grad():
            return (m(x) * cot).sum().item()

    base = L(True)
    assert base == L(False), "masks are not held fixed"
    params = [p for p in m.parameters() if p.requires_grad and p.grad is not None and p.grad.abs().max().item() > 0]
    assert len(params) > (35 if full_ft else 8)
    # direction: the gradient itself, every tensor rescaled to its parameter's scale (a random direction makes the analytic
    # value a cancellation remainder and the bf16 staircase of L dominates the comparison)
    d = [p.grad / p.grad.pow(2).mean().sqrt().clamp_min(1e-30) * p.detach().float().std().clamp_min(1e-3) for p in params]
    analytic = sum((p.grad * di).sum().item() for p, di in zip(params, d))
    ratios = {}
    for eps in (2e-3, 5e-4, 1.25e-4):
        vals = []
        for sgn in (1.0, -1.0):
            with torch.no_grad():
                for p, di in zip(params, d):
                    p.add_(di, alpha=sgn * eps)
            vals.append(L(False))
            with torch.no_grad():
                for p, di in zip(params, d):
                    p.sub_(di, alpha=sgn * eps)
        fd = (vals[0] - vals[1]) / (2 * eps)
        ratios[eps] = fd / analytic
    # measured: LoRA regime 0.978 / 1.010 / 0.988, full fine-tuning 0.647 / 0.949 / 0.9999 -- the function is strongly curved along
    # its own gradient (softmax-mean output), the finite difference converges onto the kernels' gradient as the step shrinks;
    # far below one bf16 ulp per weight the staircase averages out over the 10^5 .. 10^8 perturbed weights
    print("directional derivative, finite difference / analytic:", dict(full_ft=full_ft, exact=exact), ratios)
    assert abs(ratios[1.25e-4] - 1.0) < 0.05 and abs(ratios[5e-4] - 1.0) < 0.12, ratios
    if exact:
        assert m._engine.exact() and abs(ratios[5e-4] - 1.0) < 0.02, ratios
