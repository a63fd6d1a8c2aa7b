"""BarcodeBERT DNA encoder with LoRA -- drop-in for reference ``bioscanclip/model/dna_encoder.py``.

Keeps ``load_pre_trained_bioscan_bert``, ``get_sequence_pipeline``, ``_LoRALayer``, ``LoRA_barcode_bert`` and
``Freeze_DNA_Encoder`` with the reference's signatures and ``state_dict`` keys (dna_encoder.py:14-113).
Arithmetic (BERT-base trunk, MLM transform, replaced decoder, softmax over 768, mean over tokens) runs in
the HIP engine; there is no torch fallback.
"""
import os
from itertools import product

import torch
import torch.nn as nn
from torch import Tensor

from bioscanclip.model.arch import BertForMaskedLMParams, barcode_bert_config
from bioscanclip.model.lora import LoRAContainer


def kmer_vocab(k=5):
    """torchtext ``build_vocab_from_iterator(kmers, specials=["<MASK>","<CLS>","<UNK>"])`` restated
    (dna_encoder.py:15-17,26-28): specials take ids 0,1,2; the 4**k k-mers all have
    frequency 1 and torchtext orders ties lexicographically, which for itertools.product("ACGT") is the
    generation order, so id = 3 + sum_i d(c_i) 4**(k-1-i) with d(A,C,G,T) = (0,1,2,3)."""
    vocab = {"<MASK>": 0, "<CLS>": 1, "<UNK>": 2}
    for i, kmer in enumerate(product("ACGT", repeat=k)):
        vocab["".join(kmer)] = 3 + i
    return vocab


def get_sequence_pipeline(k=5):
    """Reference dna_encoder.py:25-35: pad/truncate to 660 nt with 'N', non-overlapping k-mers (stride k),
    unknown k-mers -> <UNK>=2, and the literal id 0 (<MASK>) prepended -> 133 ids."""
    vocab = kmer_vocab(k)
    unk = vocab["<UNK>"]
    max_len = 660

    def sequence_pipeline(x):
        x = x[:max_len] if len(x) > max_len else x + "N" * (max_len - len(x))
        toks = [x[i:i + k] for i in range(0, len(x) - k + 1, k)]
        return [0, *[vocab.get(t, unk) for t in toks]]

    return sequence_pipeline


def load_pre_trained_bioscan_bert(bioscan_bert_checkpoint, k=5):
    """Reference dna_encoder.py:14-22.  Builds the ``BertForMaskedLM(BertConfig(vocab_size=4**k+3))`` parameter
    tree and loads the BarcodeBERT checkpoint (keys with an optional ``module.`` prefix, util.py:72-84).
    ``bioscan_bert_checkpoint=None`` keeps the HF random init (synthetic benchmarks; no weights ship here)."""
    model = BertForMaskedLMParams(barcode_bert_config(k))
    if bioscan_bert_checkpoint is not None:
        if not os.path.exists(bioscan_bert_checkpoint):
            raise FileNotFoundError(bioscan_bert_checkpoint)
        state_dict = torch.load(bioscan_bert_checkpoint, map_location=torch.device("cpu"))
        state_dict = {(k_[7:] if k_.startswith("module.") else k_): v for k_, v in state_dict.items()}
        from bioscanclip.util.util import load_checked
        load_checked(model, state_dict, f"BarcodeBERT checkpoint {bioscan_bert_checkpoint}",
                     allow_unexpected=("bert.embeddings.position_ids",))   # a buffer older transformers versions saved
    return model


class _LoRALayer(nn.Module):
    """Parameter holder for ``w(x) + w_b(w_a(x))`` (reference dna_encoder.py:40-49)."""

    def __init__(self, w: nn.Module, w_a: nn.Module, w_b: nn.Module):
        super().__init__()
        self.w = w
        self.w_a = w_a
        self.w_b = w_b
        self.in_features = w.in_features

    def forward(self, x):  # pragma: no cover - guard only
        raise RuntimeError("_LoRALayer is evaluated inside the fused HIP QKV GEMM; call the LoRA_* encoder instead")


def _lora_surgery(owner, layers):
    """Wrap ``attention.self.query`` / ``.value`` of the adapted BERT layers (dna_encoder.py:73-88, language_encoder.py:56-72)."""
    for index, layer in enumerate(layers):
        if index in owner.lora_layer:
            attn = layer.attention.self
            a_q, b_q, a_v, b_v = owner._adapt(attn.query.in_features)
            attn.query = _LoRALayer(attn.query, a_q, b_q)
            attn.value = _LoRALayer(attn.value, a_v, b_v)


class LoRA_barcode_bert(LoRAContainer):
    """``LoRA_barcode_bert(model, r, num_classes, lora_layer)`` of the reference (dna_encoder.py:52-105)."""

    def __init__(self, model, r: int, num_classes: int = 0, lora_layer=None):
        super().__init__()
        layers = model.bert.encoder.layer
        # only ``None`` means "every layer" here; ``[]`` adapts none (dna_encoder.py:57-60)
        self._begin(model, r, lora_layer if lora_layer is not None else list(range(len(layers))))
        _lora_surgery(self, layers)
        self.reset_parameters()
        self.lora_barcode_bert = model
        if num_classes > 0:  # a fresh trainable decoder replaces the tied MLM decoder (dna_encoder.py:93-95)
            head = model.cls.predictions
            head.decoder = nn.Linear(head.decoder.in_features, num_classes)

    def forward(self, x: Tensor) -> Tensor:
        from bioscanclip.hip.bert_engine import barcode_bert_forward
        return barcode_bert_forward(self, x)


class Freeze_DNA_Encoder(nn.Module):
    def __init__(self):
        super(Freeze_DNA_Encoder, self).__init__()

    def forward(self, x: Tensor) -> Tensor:
        return x
