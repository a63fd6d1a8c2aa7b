"""ORACLE support -- test infrastructure only.

Deterministic synthetic weights and batches.  Every tensor is generated on the CPU from
``crc32(key) ^ seed`` with torch's (machine-independent) CPU generator, so the GPU box regenerates
bit-identical weights/inputs from the names alone and the committed fixtures (``tests/golden/``) stay
small: they hold expected *outputs* of the imported reference, never weights.
"""
import zlib

import torch


def _gen(key, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    return g


QK_GAIN = 3.0        # query / key projections: attention logits get std ~3 -> peaked, sample-dependent softmax rows
PATCH_GAIN = 4.0     # ViT patch-embed filters (made zero-sum below)
DECODER_GAIN = 8.0   # BarcodeBERT decoder: class logits std ~5 -> the softmax-mean embedding is far from uniform


_TENSOR_CACHE, _TENSOR_CACHE_BYTES, _TENSOR_CACHE_LIMIT = {}, [0], 12 << 30


def synth_tensor(key, shape, seed=0):
    """``_synth_tensor`` through a per-process cache (the tests build the same 0.2 G-parameter models dozens of times: 3 s of
    ``randn`` each).  A fresh clone is returned, so callers may train the tensors in place."""
    k = (key, tuple(shape), seed)
    t = _TENSOR_CACHE.get(k)
    if t is None:
        t = _synth_tensor(key, shape, seed)
        if t.numel() >= 4096 and _TENSOR_CACHE_BYTES[0] + t.numel() * t.element_size() <= _TENSOR_CACHE_LIMIT:
            _TENSOR_CACHE[k] = t
            _TENSOR_CACHE_BYTES[0] += t.numel() * t.element_size()
            return t.clone()
        return t
    return t.clone()


def _synth_tensor(key, shape, seed=0):
    """Value rule chosen so that activations and LoRA branches are O(0.1-1) and nothing on the path is degenerate:
    LN gains ~1, biases small, Linear weights ~0.6/sqrt(fan_in), LoRA B non-zero; the query and key
    projections are QK_GAIN times larger so attention rows are peaked (with 0.6/sqrt(fan_in) everywhere the scores have std
    0.36, every softmax row is uniform and the Q-LoRA gradient is a cancellation remainder), and the BarcodeBERT decoder is
    DECODER_GAIN times larger so softmax(logits) is not the uniform vector (else every DNA embedding is the same direction,
    every similarity is 1 and the contrastive loss is exactly ln N whatever the encoders do)."""
    g = _gen(key, seed)
    shape = tuple(shape)
    r = lambda s: torch.randn(shape, generator=g, dtype=torch.float32) * s
    leaf = key.rsplit(".", 1)[-1]
    is_ln = ("LayerNorm." in key) or (".norm1." in key) or (".norm2." in key) or key.endswith("norm.weight") \
        or key.endswith("norm.bias")
    if is_ln:
        return 1.0 + r(0.1) if leaf == "weight" else r(0.05)
    if "embeddings." in key:
        return r(0.05)
    if key.endswith("cls_token") or key.endswith("pos_embed"):
        return r(0.02)
    if ".linear_a_" in key or ".w_a." in key:
        return r(0.03)
    if ".linear_b_" in key or ".w_b." in key:
        return r(0.02)
    if leaf == "bias":
        return r(0.02)
    if leaf == "weight" and len(shape) >= 2:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        w = r(0.6 / fan_in ** 0.5)
        if "patch_embed.proj." in key:
            # band-pass filters (zero sum over the 3x16x16 window), as trained first-layer filters mostly are: with a DC
            # gain every patch of every [0,1] image maps to the same dominant vector and all image embeddings coincide
            w = (w - w.mean(dim=tuple(range(1, len(shape))), keepdim=True)) * PATCH_GAIN
        elif ".attn.qkv." in key and shape[0] % 3 == 0:        # timm fused qkv: rows [q | k | v]
            w[: 2 * shape[0] // 3] *= QK_GAIN
        elif ".attention.self.query." in key or ".attention.self.key." in key:   # HF BERT (LoRA-wrapped or plain)
            w *= QK_GAIN
        elif ".cls.predictions.decoder." in key:
            w *= DECODER_GAIN
        return w
    return r(0.02)


def synth_state_dict(shapes, seed=0):
    """``shapes``: mapping key -> shape (e.g. from ``module.state_dict()``)."""
    return {k: synth_tensor(k, s, seed) for k, s in shapes.items()}


def shapes_of(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def synth_batch(batch, seed=0, with_text=False, image_size=224, dna_tokens=133, text_tokens=20,
                dna_vocab=1027, text_vocab=30522, dup_labels=False, label_offset=0):
    """Synthetic batch in the reference's layout: image fp32 in [0,1] (no mean/std),
    DNA ids = [0, 132 ids in [3, vocab)], text ids with [CLS]=101 ... [SEP]=102 + zero padding and a matching
    attention mask, identity labels (or ~10 % duplicated)."""
    g = _gen("batch", seed)
    # image: values in [0,1] (reference range after ToTensor, no mean/std).  Each image = its own low-frequency pattern
    # (three plane waves per channel, <= 4 cycles across the image) + uniform pixel noise, so different images differ in
    # every patch and the encoders give distinct embeddings (i.i.d. uniform pixels average out to the same vector).
    noise = torch.rand((batch, 3, image_size, image_size), generator=g, dtype=torch.float32)
    ax = torch.arange(image_size, dtype=torch.float32) / image_size
    fx = torch.randint(0, 5, (batch, 3, 3), generator=g).to(torch.float32)
    fy = torch.randint(0, 5, (batch, 3, 3), generator=g).to(torch.float32)
    ph = torch.rand((batch, 3, 3), generator=g, dtype=torch.float32) * 6.283185307179586
    am = torch.rand((batch, 3, 3), generator=g, dtype=torch.float32)
    arg = 6.283185307179586 * (fx[..., None, None] * ax[None, None, None, None, :] + fy[..., None, None]
                               * ax[None, None, None, :, None]) + ph[..., None, None]
    pattern = (am[..., None, None] * torch.cos(arg)).sum(2) / 3.0
    image = (0.5 + 1.2 * pattern + 0.3 * (noise - 0.5)).clamp_(0.0, 1.0)
    # DNA: every barcode draws 80 % of its k-mers from its own pool of 12 (a sample-specific composition, as real barcodes
    # of one species share motifs), the rest uniformly; id_0 = 0 (<MASK>, dna_encoder.py:33).
    pool = torch.randint(3, dna_vocab, (batch, 12), generator=g, dtype=torch.int64)
    pick = torch.randint(0, 12, (batch, dna_tokens), generator=g)
    dna = torch.gather(pool, 1, pick)
    rnd = torch.randint(3, dna_vocab, (batch, dna_tokens), generator=g, dtype=torch.int64)
    use_rnd = torch.rand((batch, dna_tokens), generator=g) < 0.2
    dna = torch.where(use_rnd, rnd, dna)
    dna[:, 0] = 0
    label = torch.arange(batch, dtype=torch.int64) + label_offset
    if dup_labels and batch >= 4:
        src = torch.randint(0, batch, (max(1, batch // 10) * 2,), generator=g)
        label[src[1::2]] = label[src[0::2]]
    text = None
    if with_text:
        ids = torch.randint(1000, text_vocab, (batch, text_tokens), generator=g, dtype=torch.int64)
        lens = torch.randint(6, text_tokens + 1, (batch,), generator=g)
        pos = torch.arange(text_tokens)[None, :]
        mask = (pos < lens[:, None]).to(torch.int64)
        ids = ids * mask
        ids[:, 0] = 101
        ids[torch.arange(batch), lens - 1] = 102
        text = {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask}
    return image, dna, text, label
