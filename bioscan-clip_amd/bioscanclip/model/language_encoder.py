"""BERT-small text encoder with LoRA -- drop-in for reference ``bioscanclip/model/language_encoder.py``.

``load_pre_trained_bert``, ``_LoRALayer`` and ``LoRA_bert`` keep the reference's signatures and ``state_dict``
keys (language_encoder.py:12-89).  Arithmetic (4-layer BERT trunk with additive key mask, unmasked mean over
the 20 positions, ``proj`` Linear(512, 768)) runs in the HIP engine; there is no torch fallback.
"""
import math

import torch
import torch.nn as nn
from torch import Tensor

from bioscanclip.model.arch import BertModelParams, bert_small_config
from bioscanclip.model.dna_encoder import _LoRALayer, _lora_surgery  # noqa: F401  (same class in the reference)


def load_pre_trained_bert(checkpoint=None):
    """Reference language_encoder.py:12-20 fetches tokenizer + weights of "prajjwal1/bert-small" from the hub.
    There is no network here: this builds the same parameter tree (hidden 512, 4 layers, 8 heads, FFN 2048,
    vocab 30522 -- public model card), optionally loads a local ``state_dict`` file, freezes
    it, and returns ``(None, model)`` -- the tokenizer is never used on the training path (tokens come
    pre-computed from the HDF5 file)."""
    model = BertModelParams(bert_small_config())
    if checkpoint is not None:
        model.load_state_dict(torch.load(checkpoint, map_location="cpu"), strict=False)
    for param in model.parameters():
        param.requires_grad = False
    return None, model


class LoRA_bert(nn.Module):
    def __init__(self, model, r: int, num_classes: int = 0, lora_layer=None):
        super(LoRA_bert, self).__init__()

        assert r > 0
        self.r = r
        if lora_layer is not None:
            self.lora_layer = lora_layer
        else:
            self.lora_layer = list(range(len(model.encoder.layer)))

        self.w_As = []
        self.w_Bs = []

        for param in model.parameters():
            param.requires_grad = False

        _lora_surgery(self, model.encoder.layer, r, self.lora_layer)
        self.reset_parameters()
        self.lora_bert = model

        if num_classes > 0:
            self.proj = nn.Linear(self.lora_bert.pooler.dense.out_features, num_classes)
        self._engine = None

    def reset_parameters(self) -> None:
        for w_A in self.w_As:
            nn.init.kaiming_uniform_(w_A.weight, a=math.sqrt(5))
        for w_B in self.w_Bs:
            nn.init.zeros_(w_B.weight)

    def _load_from_state_dict(self, *args, **kwargs):
        self._engine = None
        return super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x) -> Tensor:
        from bioscanclip.hip.bert_engine import bert_text_forward
        return bert_text_forward(self, x)
