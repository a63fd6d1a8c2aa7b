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


def synth_tensor(key, shape, seed=0):
    """Value rule chosen so that activations, attention logits and LoRA branches are all O(0.1-1):
    LN gains ~1, biases small, Linear weights ~0.6/sqrt(fan_in), LoRA B non-zero."""
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
        return r(0.6 / fan_in ** 0.5)
    return r(0.02)


def synth_state_dict(shapes, seed=0):
    """``shapes``: mapping key -> shape (e.g. from ``module.state_dict()``)."""
    return {k: synth_tensor(k, s, seed) for k, s in shapes.items()}


def shapes_of(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def synth_batch(batch, seed=0, with_text=False, image_size=224, dna_tokens=133, text_tokens=20,
                dna_vocab=1027, text_vocab=30522, dup_labels=False, label_offset=0):
    """Synthetic batch in the reference's layout: image uniform [0,1) fp32 (no mean/std),
    DNA ids = [0, 132 ids in [3, vocab)], text ids with [CLS]=101 ... [SEP]=102 + zero padding and a matching
    attention mask, identity labels (or ~10 % duplicated)."""
    g = _gen("batch", seed)
    image = torch.rand((batch, 3, image_size, image_size), generator=g, dtype=torch.float32)
    dna = torch.randint(3, dna_vocab, (batch, dna_tokens), generator=g, dtype=torch.int64)
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
