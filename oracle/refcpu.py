"""ORACLE -- test infrastructure only, never the product path.

CPU restatement (pure torch, fp32) of the one hot path this repo accelerates: the BIOSCAN-CLIP
contrastive training step.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package; the product (``bioscan-clip_amd/``) never does.

Every function works on a flat ``state_dict`` (reference key names) so the same
weights can be fed to (a) the imported reference modules (``oracle/gen_golden.py``, this container
only), (b) this restatement, (c) the HIP engine.

Pinning: ``oracle/gen_golden.py`` runs the *imported reference* (`/root/reference`) on seeded inputs
and stores its outputs under ``tests/golden/``; ``tests/test_01_oracle_golden.py`` checks this file
against those fixtures.  The third-party arithmetic the reference delegates to (timm 0.6.13 ViT,
HF transformers 4.29.2 BERT) is restated from its published semantics:
the BERT half is pinned through the HF classes the reference really calls (transformers 5.15 here),
the ViT half only through the reference's own LoRA wrapper around a timm-shaped module, because timm
itself is absent from this image -- "parity unpinned" for timm's internals, as DESIGN.md records.
"""
import math

import torch


# Rounding sites left exact under emulate_bf16 (tools/site_sensitivity.py: which operand roundings carry the distance to the f32
# reference; hip/engine.py's BSCLIP_PARITY=3 splits exactly the sites named here).  Site names: "qkv.a" / "qkv.w" (QKV GEMM
# operands), "q" / "k" / "v" (the attention operands = the QKV GEMM's stored output), "p" (un-normalised probabilities),
# "proj.a" / "proj.w", "fc1.a" / "fc1.w", "fc2.a" / "fc2.w", "resid", "head.a" / "head.w", "lora" (t and the LoRA-B operand).
EXACT_SITES = set()


def _q(x, emulate_bf16, site=None):
    """Optional bf16 rounding of a GEMM operand (straight-through for autograd).  Used to restate
    *where* the HIP path rounds (A/B operands of every MFMA GEMM) while keeping fp32 math."""
    if not emulate_bf16 or (site is not None and site in EXACT_SITES):
        return x
    return x + (x.detach().to(torch.bfloat16).to(x.dtype) - x.detach())


# The HIP engines keep the residual stream in bf16 by default (hip/engine.py RESID_STREAM_BF16): every sub-layer's sum is formed
# in f32 and rounded once when stored.  The rounding-aware evaluation (emulate_bf16=True) restates that rounding point too; set
# to False to emulate the f32 stream (BSCLIP_RESID_STREAM=f32).  The plain f32 evaluation never rounds anything.
EMULATE_RESID_BF16 = True
# The ViT patch embedding runs on split-bf16 operands (hip/engine.py PATCH_SPLIT: hi.hi + lo.hi + hi.lo, ~2^-16 relative): the
# rounding-aware evaluation leaves that one GEMM's operands unrounded.  False emulates the plain bf16 GEMM (BSCLIP_PATCH_SPLIT=0).
EMULATE_PATCH_SPLIT = True


def _rq(x, emulate_bf16):
    """bf16 rounding of a stored residual-stream tensor (only under emulate_bf16)."""
    return _q(x, emulate_bf16 and EMULATE_RESID_BF16, "resid")


def _q8(x, scale=None):
    """OCP fp8 e4m3 rounding as the fp8 trunks apply it (BASELINE configs[4]; hip/engine.py _pack_fp8, csrc/norm.hip Y_FP8,
    csrc/gemm.hip EPI_GELU_FP8): activations with scale 1, saturated at +-448; weights with one scale per output row (row amax
    -> 448).  Straight-through for autograd, like ``_q``."""
    d = x.detach()
    if scale is None:
        r = d.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(x.dtype)
    else:
        r = (d / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(x.dtype) * scale
    return x + (r - d)


def linear_fp8(x, w, b=None):
    """y = e4m3(x) . e4m3_rowscaled(w)^T + b in f32: the forward GEMMs of the fp8 trunks (QKV, fc1, fc2)."""
    sc = (w.detach().abs().amax(dim=1, keepdim=True) / 448.0).clamp_min(1e-30)
    y = _q8(x) @ _q8(w, sc).t()
    return y if b is None else y + b


def linear(x, w, b=None, emulate_bf16=False, site=None):
    y = _q(x, emulate_bf16, site and site + ".a") @ _q(w, emulate_bf16, site and site + ".w").t()
    return y if b is None else y + b


def gelu_erf(x):
    """Exact GELU (timm ``nn.GELU``; HF ``hidden_act='gelu'``)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def layer_norm(x, w, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def sdpa(q, k, v, bias=None, emulate_bf16=False):
    """softmax(q k^T / sqrt(d) + bias) v over [B, heads, S, d].

    ``emulate_bf16`` restates the one rounding point inside the HIP attention kernel that is not a GEMM operand of a
    Linear: the un-normalised probabilities exp(s - max) are rounded to bf16 before the P.V product, while the row
    sum that normalises the result is taken over the unrounded f32 values (csrc/attn.hip, attn_fwd_kernel)."""
    s = (q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    if bias is not None:
        s = s + bias
    if not emulate_bf16:
        return torch.softmax(s, dim=-1) @ v
    e = torch.exp(s - s.amax(dim=-1, keepdim=True))
    return (_q(e, True, "p") @ v) / e.sum(dim=-1, keepdim=True)


# --------------------------------------------------------------------------------------
# ViT-B/16 + LoRA  (reference image_encoder.py:15-48, 51-109; timm 0.6.13 semantics App. A.1)
# --------------------------------------------------------------------------------------
def vit_encoder(sd, image, prefix="image_encoder.lora_vit.", num_heads=12, emulate_bf16=False,
                return_hidden=False, taps=None, emulate_fp8=False):
    """``LoRA_ViT_timm.forward`` (image_encoder.py:108-109) -> timm ``VisionTransformer.forward``:
    patch-embed conv (k=s=16) -> cat cls -> +pos -> pre-LN blocks (eps 1e-6) with LoRA added in place to
    the Q and V slices of the fused qkv output, scale 1 (image_encoder.py:42-48) -> norm -> token 0 -> head."""
    p = lambda k: sd[prefix + k]
    eb = emulate_bf16 or emulate_fp8
    # fp8 trunks (emulate_fp8): QKV / fc1 / fc2 take e4m3 operands, everything else rounds as the bf16 path does; the fp8 engines
    # keep the f32 residual stream
    rq = (lambda t, e: t) if emulate_fp8 else _rq
    lin8 = (lambda x, w, b, e, site=None: linear_fp8(x, w, b)) if emulate_fp8 else linear
    B = image.shape[0]
    w_pe = p("patch_embed.proj.weight")
    D = w_pe.shape[0]
    ps = w_pe.shape[-1]
    # conv k=s=16 == im2col GEMM; column order (c, ky, kx) matches weight.flatten(1)
    gh, gw = image.shape[2] // ps, image.shape[3] // ps
    cols = image.reshape(B, 3, gh, ps, gw, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, 3 * ps * ps)
    x = linear(cols, w_pe.reshape(D, -1), p("patch_embed.proj.bias"), eb and not EMULATE_PATCH_SPLIT)
    x = rq(torch.cat([p("cls_token").expand(B, -1, -1), x], dim=1) + p("pos_embed"), eb)
    tap = (lambda name, t: taps.__setitem__(name, t.detach())) if taps is not None else (lambda name, t: None)
    tap("x0", x)
    depth = 0
    while (prefix + f"blocks.{depth}.norm1.weight") in sd:
        depth += 1
    hd = D // num_heads
    for i in range(depth):
        b = f"blocks.{i}."
        h = layer_norm(x, p(b + "norm1.weight"), p(b + "norm1.bias"), 1e-6)
        if (prefix + b + "attn.qkv.qkv.weight") in sd:  # _LoRA_qkv_timm surgery applied
            qkv = lin8(h, p(b + "attn.qkv.qkv.weight"), p(b + "attn.qkv.qkv.bias"), eb, site="qkv")
            # emulation note: the HIP LayerNorm kernel forms t = y A^T from the f32 row and the f32 master A and rounds
            # only t (csrc/norm.hip), so the inner product is NOT taken on rounded operands
            t_q = linear(h, p(b + "attn.qkv.linear_a_q.weight"))
            t_v = linear(h, p(b + "attn.qkv.linear_a_v.weight"))
            tap(f"t.{i}", torch.cat([t_q, t_v], dim=-1))
            new_q = linear(t_q, p(b + "attn.qkv.linear_b_q.weight"), None, eb, site="lora")
            new_v = linear(t_v, p(b + "attn.qkv.linear_b_v.weight"), None, eb, site="lora")
            qkv = torch.cat([qkv[..., :D] + new_q, qkv[..., D:2 * D], qkv[..., 2 * D:] + new_v], dim=-1)
        else:
            qkv = lin8(h, p(b + "attn.qkv.weight"), p(b + "attn.qkv.bias"), eb, site="qkv")
        S = qkv.shape[1]
        qkv = qkv.reshape(B, S, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
        tap(f"h1.{i}", h)
        tap(f"qkv.{i}", qkv.permute(1, 3, 0, 2, 4).reshape(B, S, 3 * D))
        ctx = sdpa(_q(qkv[0], eb, "q"), _q(qkv[1], eb, "k"), _q(qkv[2], eb, "v"), emulate_bf16=eb).transpose(1, 2).reshape(B, S, D)
        tap(f"ctx.{i}", ctx)
        x = rq(x + linear(ctx, p(b + "attn.proj.weight"), p(b + "attn.proj.bias"), eb, site="proj"), eb)
        tap(f"x{2 * i + 1}", x)
        h = layer_norm(x, p(b + "norm2.weight"), p(b + "norm2.bias"), 1e-6)
        h = gelu_erf(lin8(h, p(b + "mlp.fc1.weight"), p(b + "mlp.fc1.bias"), eb, site="fc1"))
        x = rq(x + lin8(h, p(b + "mlp.fc2.weight"), p(b + "mlp.fc2.bias"), eb, site="fc2"), eb)
        tap(f"x{2 * i + 2}", x)
    x = layer_norm(x, p("norm.weight"), p("norm.bias"), 1e-6)
    if return_hidden:
        return x
    return linear(x[:, 0], p("head.weight"), p("head.bias"), eb, site="head")


# --------------------------------------------------------------------------------------
# HF BERT encoder with LoRA on query/value (reference dna_encoder.py:40-49,73-88; language_encoder.py:24-33)
# --------------------------------------------------------------------------------------
def _lora_or_plain(sd, base, h, eb, f8=False):
    """``_LoRALayer.forward``: ``w(x) + w_b(w_a(x))`` (dna_encoder.py:47-49) or the untouched Linear."""
    lin = (lambda x, w, b, e, site=None: linear_fp8(x, w, b)) if f8 else linear
    if (base + "w.weight") in sd:
        # (emulation: t = h w_a^T is formed in f32 and only t is rounded, as in the ViT branch above; fp8 trunks: the frozen
        # weight takes e4m3 operands, the LoRA branch stays bf16)
        return lin(h, sd[base + "w.weight"], sd[base + "w.bias"], eb, site="qkv") + linear(
            linear(h, sd[base + "w_a.weight"]), sd[base + "w_b.weight"], None, eb, site="lora")
    return lin(h, sd[base + "weight"], sd[base + "bias"], eb, site="qkv")


def bert_encoder(sd, prefix, input_ids, token_type_ids=None, attention_mask=None, num_heads=12, eps=1e-12,
                 emulate_bf16=False, dropout_p=0.0, emulate_fp8=False):
    """HF ``BertModel`` trunk (post-LN, abs positions; App. A.2): embeddings LN(word+pos+type) then layers
    ``h1 = LN(h + dense(ctx))``, ``h2 = LN(h1 + dense2(gelu(dense1(h1))))``.  ``prefix`` ends before
    ``embeddings.``.  Dropout (HF p=0.1 in train mode) is only restated for p=0: parity runs disable it."""
    assert dropout_p == 0.0, "oracle restates the deterministic (p=0) path; dropout is tested statistically"
    eb = emulate_bf16 or emulate_fp8
    f8 = emulate_fp8
    rq = (lambda t, e: t) if f8 else _rq          # the fp8 engines keep the f32 residual stream
    lin8 = (lambda x, w, b, e, site=None: linear_fp8(x, w, b)) if f8 else linear
    B, S = input_ids.shape
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    pos = torch.arange(S)
    e = "embeddings."
    # HF BertEmbeddings: nn.Embedding(vocab, hidden, padding_idx=config.pad_token_id = 0) -- row 0 receives no gradient
    # (matters for full fine-tuning only: every barcode starts with id 0, get_sequence_pipeline's literal <MASK>)
    h = torch.nn.functional.embedding(input_ids, sd[prefix + e + "word_embeddings.weight"], padding_idx=0) \
        + sd[prefix + e + "token_type_embeddings.weight"][token_type_ids] \
        + sd[prefix + e + "position_embeddings.weight"][pos][None]
    h = layer_norm(h, sd[prefix + e + "LayerNorm.weight"], sd[prefix + e + "LayerNorm.bias"], eps)
    bias = None
    if attention_mask is not None:
        # HF extended mask: (1 - m) * finfo.min added to the scores of padded keys (App. A.3)
        bias = (1.0 - attention_mask[:, None, None, :].to(h.dtype)) * torch.finfo(h.dtype).min
    H = h.shape[-1]
    hd = H // num_heads
    L = 0
    while (prefix + f"encoder.layer.{L}.attention.output.dense.weight") in sd:
        L += 1
    for i in range(L):
        lp = prefix + f"encoder.layer.{i}."
        q = _lora_or_plain(sd, lp + "attention.self.query.", h, eb, f8)
        k = _lora_or_plain(sd, lp + "attention.self.key.", h, eb, f8)
        v = _lora_or_plain(sd, lp + "attention.self.value.", h, eb, f8)
        sh = lambda t: t.reshape(B, S, num_heads, hd).transpose(1, 2)
        ctx = sdpa(_q(sh(q), eb, "q"), _q(sh(k), eb, "k"), _q(sh(v), eb, "v"), bias, emulate_bf16=eb).transpose(1, 2).reshape(B, S, H)
        a = linear(ctx, sd[lp + "attention.output.dense.weight"], sd[lp + "attention.output.dense.bias"], eb, site="proj")
        # (bf16 stream: the LayerNorm's bf16 GEMM operand IS the residual branch, and the sum is stored rounded)
        h = layer_norm(rq(rq(h, eb) + a, eb), sd[lp + "attention.output.LayerNorm.weight"],
                       sd[lp + "attention.output.LayerNorm.bias"], eps)
        m = gelu_erf(lin8(h, sd[lp + "intermediate.dense.weight"], sd[lp + "intermediate.dense.bias"], eb, site="fc1"))
        m = lin8(m, sd[lp + "output.dense.weight"], sd[lp + "output.dense.bias"], eb, site="fc2")
        h = layer_norm(rq(rq(h, eb) + m, eb), sd[lp + "output.LayerNorm.weight"], sd[lp + "output.LayerNorm.bias"], eps)
    return h


def barcode_bert_encoder(sd, ids, prefix="dna_encoder.lora_barcode_bert.", num_heads=12, emulate_bf16=False, emulate_fp8=False):
    """``LoRA_barcode_bert.forward`` (dna_encoder.py:103-105): ``BertForMaskedLM(x).logits.softmax(-1).mean(1)``
    with only ``input_ids`` passed (no mask, token_type 0) and ``cls.predictions.decoder`` replaced by a fresh
    Linear(768, num_classes) with its own bias (dna_encoder.py:93-95)."""
    eb = emulate_bf16 or emulate_fp8
    h = bert_encoder(sd, prefix + "bert.", ids, num_heads=num_heads, emulate_bf16=eb, emulate_fp8=emulate_fp8)
    t = prefix + "cls.predictions."
    h = gelu_erf(linear(h, sd[t + "transform.dense.weight"], sd[t + "transform.dense.bias"], eb, site="head"))
    h = layer_norm(h, sd[t + "transform.LayerNorm.weight"], sd[t + "transform.LayerNorm.bias"], 1e-12)
    logits = linear(h, sd[t + "decoder.weight"], sd[t + "decoder.bias"], eb, site="head")
    return torch.softmax(logits, dim=-1).mean(dim=1)


def bert_text_encoder(sd, language_input, prefix="language_encoder.", num_heads=8, emulate_bf16=False):
    """``LoRA_bert.forward`` (language_encoder.py:87-89): ``proj(BertModel(**x).last_hidden_state.mean(1))`` --
    the mean runs over all positions, padding included (App. A.3 / B-7)."""
    h = bert_encoder(sd, prefix + "lora_bert.", language_input["input_ids"],
                     language_input.get("token_type_ids"), language_input.get("attention_mask"),
                     num_heads=num_heads, emulate_bf16=emulate_bf16)
    return linear(h.mean(dim=1), sd[prefix + "proj.weight"], sd[prefix + "proj.bias"], emulate_bf16, site="head")


def l2_normalize(x, eps=1e-12):
    """``F.normalize(x, p=2, dim=-1)``: x / max(||x||, eps)."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def simple_clip_forward(sd, image, dna, language, emulate_bf16=False, vit_heads=12, dna_heads=12, txt_heads=8):
    """``SimpleCLIP.forward`` (simple_clip.py:27-50): DNA, image, text encoders, each L2-normalised;
    ``None`` for an absent modality."""
    img = dna_o = txt = None
    if dna is not None:
        dna_o = l2_normalize(barcode_bert_encoder(sd, dna, num_heads=dna_heads, emulate_bf16=emulate_bf16))
    if image is not None:
        img = l2_normalize(vit_encoder(sd, image, num_heads=vit_heads, emulate_bf16=emulate_bf16))
    if language is not None:
        txt = l2_normalize(bert_text_encoder(sd, language, num_heads=txt_heads, emulate_bf16=emulate_bf16))
    return img, dna_o, txt


# --------------------------------------------------------------------------------------
# loss (reference loss_func.py:18-54) -- App. A.6
# --------------------------------------------------------------------------------------
def construct_label_matrix(labels):
    """``construct_label_metrix`` (loss_func.py:18-21)."""
    return (labels.unsqueeze(0) == labels.unsqueeze(1)).float()


def soft_target_ce(logits, target):
    """``nn.CrossEntropyLoss()`` with float (probability-style) targets, mean over rows; target rows are NOT
    normalised (App. A.6)."""
    return -(target * torch.log_softmax(logits, dim=1)).sum(dim=1).mean()


def contrastive_loss(image_features, dna_features, text_features, label, logit_scale=1.0 / 0.07):
    """``ContrastiveLoss.forward`` (loss_func.py:29-54), term by term (every ordered pair contributes both
    directions, so each distinct matrix is counted twice; the mean is unchanged)."""
    feats = [f for f in (image_features, dna_features, text_features) if f is not None]
    if len(feats) < 2:
        raise ValueError("Too less element for calculating the contrastive loss.")
    T = construct_label_matrix(label)
    terms = []
    for ia, fa in enumerate(feats):
        for ib, fb in enumerate(feats):
            if ia == ib:
                continue
            a = l2_normalize(fa)
            b = l2_normalize(fb)
            terms.append(soft_target_ce(logit_scale * a @ b.t(), T))
            terms.append(soft_target_ce(logit_scale * b @ a.t(), T))
    return sum(terms) * 1.0 / len(terms)


# --------------------------------------------------------------------------------------
# AdamW (torch.optim.AdamW defaults used at train_cl.py:158) and the whole step (train_epoch.py:21-44)
# --------------------------------------------------------------------------------------
def adamw_update(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01):
    """One decoupled-weight-decay Adam update, in place; ``step`` is 1-based."""
    p.mul_(1.0 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def is_trainable_key(k):
    """Trainable set in the LoRA regime (App. A.5): LoRA A/B, ViT head, DNA decoder, text proj."""
    return (".linear_a_" in k or ".linear_b_" in k or ".w_a." in k or ".w_b." in k
            or k.startswith("image_encoder.lora_vit.head.")
            or k.startswith("dna_encoder.lora_barcode_bert.cls.predictions.decoder.")
            or k.startswith("language_encoder.proj."))


class StepState:
    """Trainable leaf tensors + AdamW moments for :func:`train_step`."""

    def __init__(self, sd):
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.train_keys = [k for k in self.sd if is_trainable_key(k) and self.sd[k].is_floating_point()]
        for k in self.train_keys:
            self.sd[k].requires_grad_(True)
        self.m = {k: torch.zeros_like(self.sd[k]) for k in self.train_keys}
        self.v = {k: torch.zeros_like(self.sd[k]) for k in self.train_keys}
        self.step = 0


def loss_and_grads(state, image, dna, language, label, emulate_bf16=False, logit_scale=1.0 / 0.07):
    for k in state.train_keys:
        state.sd[k].grad = None
    img, dn, txt = simple_clip_forward(state.sd, image, dna, language, emulate_bf16=emulate_bf16)
    loss = contrastive_loss(img, dn, txt, label, logit_scale)
    loss.backward()
    grads = {k: state.sd[k].grad for k in state.train_keys}
    return loss.detach(), (img, dn, txt), grads


def train_step(state, image, dna, language, label, lr=1e-3, emulate_bf16=False):
    """One iteration of the reference loop body (train_epoch.py:28-42): zero_grad, forward, loss, backward,
    AdamW step (defaults; weight decay also on biases, App. B-4)."""
    loss, outs, grads = loss_and_grads(state, image, dna, language, label, emulate_bf16)
    state.step += 1
    with torch.no_grad():
        for k in state.train_keys:
            if grads[k] is None:
                continue
            adamw_update(state.sd[k], grads[k], state.m[k], state.v[k], state.step, lr)
    return loss, outs, grads
