"""BarcodeBERT DNA encoder with LoRA -- drop-in for reference ``bioscanclip/model/dna_encoder.py``.

Keeps ``load_pre_trained_bioscan_bert``, ``get_sequence_pipeline``, ``_LoRALayer``, ``LoRA_barcode_bert`` and
``Freeze_DNA_Encoder`` with the reference's signatures and ``state_dict`` keys (dna_encoder.py:14-113).
Arithmetic (BERT-base trunk, MLM transform, replaced decoder, softmax over 768, mean over tokens) runs in
the HIP engine; there is no torch fallback.
"""
import math
import os
from itertools import product

import torch
import torch.nn as nn
from torch import Tensor

from bioscanclip.model.arch import BertForMaskedLMParams, barcode_bert_config


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
        model.load_state_dict(state_dict, strict=False)
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


def _lora_surgery(owner, layers, r, lora_layer):
    """Shared by ``LoRA_barcode_bert`` / ``LoRA_bert`` (dna_encoder.py:73-88, language_encoder.py:56-72)."""
    for layer_idx, layer in enumerate(layers):
        if layer_idx not in lora_layer:
            continue
        w_q_linear = layer.attention.self.query
        w_v_linear = layer.attention.self.value
        dim = layer.attention.self.query.in_features
        w_a_linear_q = nn.Linear(dim, r, bias=False)
        w_b_linear_q = nn.Linear(r, dim, bias=False)
        w_a_linear_v = nn.Linear(dim, r, bias=False)
        w_b_linear_v = nn.Linear(r, dim, bias=False)
        owner.w_As.append(w_a_linear_q)
        owner.w_Bs.append(w_b_linear_q)
        owner.w_As.append(w_a_linear_v)
        owner.w_Bs.append(w_b_linear_v)
        layer.attention.self.query = _LoRALayer(w_q_linear, w_a_linear_q, w_b_linear_q)
        layer.attention.self.value = _LoRALayer(w_v_linear, w_a_linear_v, w_b_linear_v)


class LoRA_barcode_bert(nn.Module):
    def __init__(self, model, r: int, num_classes: int = 0, lora_layer=None):
        super(LoRA_barcode_bert, self).__init__()

        assert r > 0
        self.r = r
        # reference dna_encoder.py:57-60 -- ``is not None`` so ``[]`` means "no LoRA" 
        if lora_layer is not None:
            self.lora_layer = lora_layer
        else:
            self.lora_layer = list(range(len(model.bert.encoder.layer)))

        self.w_As = []
        self.w_Bs = []

        for param in model.parameters():
            param.requires_grad = False

        _lora_surgery(self, model.bert.encoder.layer, r, self.lora_layer)
        self.reset_parameters()
        self.lora_barcode_bert = model

        if num_classes > 0:
            self.lora_barcode_bert.cls.predictions.decoder = nn.Linear(
                self.lora_barcode_bert.cls.predictions.decoder.in_features, num_classes)
        self._engine = None

    def reset_parameters(self) -> None:
        for w_A in self.w_As:
            nn.init.kaiming_uniform_(w_A.weight, a=math.sqrt(5))
        for w_B in self.w_Bs:
            nn.init.zeros_(w_B.weight)

    def _load_from_state_dict(self, *args, **kwargs):
        self._engine = None
        return super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x: Tensor) -> Tensor:
        from bioscanclip.hip.bert_engine import barcode_bert_forward
        return barcode_bert_forward(self, x)


class Freeze_DNA_Encoder(nn.Module):
    def __init__(self):
        super(Freeze_DNA_Encoder, self).__init__()

    def forward(self, x: Tensor) -> Tensor:
        return x
