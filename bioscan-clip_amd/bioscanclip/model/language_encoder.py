"""BERT-small text encoder with LoRA -- drop-in for reference ``bioscanclip/model/language_encoder.py``.

``load_pre_trained_bert``, ``_LoRALayer`` and ``LoRA_bert`` keep the reference's signatures and ``state_dict``
keys (language_encoder.py:12-89).  Arithmetic (4-layer BERT trunk with additive key mask, unmasked mean over
the 20 positions, ``proj`` Linear(512, 768)) runs in the HIP engine; there is no torch fallback.
"""

import torch
import torch.nn as nn
from torch import Tensor

from bioscanclip.model.arch import BertModelParams, bert_small_config
from bioscanclip.model.dna_encoder import _LoRALayer, _lora_surgery  # noqa: F401  (same class in the reference)
from bioscanclip.model.lora import LoRAContainer


def load_pre_trained_bert(checkpoint=None):
    """Reference language_encoder.py:12-20 fetches tokenizer + weights of "prajjwal1/bert-small" from the hub.
    There is no network here: this builds the same parameter tree (hidden 512, 4 layers, 8 heads, FFN 2048,
    vocab 30522 -- public model card), optionally loads a local ``state_dict`` file, freezes
    it, and returns ``(None, model)`` -- the tokenizer is never used on the training path (tokens come
    pre-computed from the HDF5 file)."""
    model = BertModelParams(bert_small_config())
    if checkpoint is not None:
        from bioscanclip.util.util import load_checked
        load_checked(model, torch.load(checkpoint, map_location="cpu"), f"BERT-small checkpoint {checkpoint}",
                     allow_unexpected=("embeddings.position_ids",))
    for param in model.parameters():
        param.requires_grad = False
    return None, model


class LoRA_bert(LoRAContainer):
    """``LoRA_bert(model, r, num_classes, lora_layer)`` of the reference (language_encoder.py:36-89)."""

    def __init__(self, model, r: int, num_classes: int = 0, lora_layer=None):
        super().__init__()
        layers = model.encoder.layer
        self._begin(model, r, lora_layer if lora_layer is not None else list(range(len(layers))))
        _lora_surgery(self, layers)
        self.reset_parameters()
        self.lora_bert = model
        if num_classes > 0:  # trainable projection of the token mean into the shared space (language_encoder.py:76-77)
            self.proj = nn.Linear(model.pooler.dense.out_features, num_classes)

    def forward(self, x) -> Tensor:
        from bioscanclip.hip.bert_engine import bert_text_forward
        return bert_text_forward(self, x)
