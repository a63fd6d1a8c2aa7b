"""LoRA-adapted ViT-B/16 image encoder -- drop-in for reference ``bioscanclip/model/image_encoder.py``.

Same classes, constructor signatures, attribute names and ``state_dict`` keys as the reference
(``_LoRA_qkv_timm`` image_encoder.py:15-48, ``LoRA_ViT_timm`` :51-109), but the modules hold parameters
only: ``forward`` dispatches the whole encoder (patch-embed -> 12 pre-LN blocks with the rank-r LoRA
branch folded into the QKV GEMM's K dimension -> LN -> token 0 -> head) to the hand-written HIP engine
(``bioscanclip.hip``).  There is no torch fallback: on a machine without the built HIP library the call
raises.
"""
import torch.nn as nn
from torch import Tensor

from bioscanclip.model.lora import LoRAContainer


class _LoRA_qkv_timm(nn.Module):
    """Parameter holder with the reference's names (``qkv, linear_a_q, linear_b_q, linear_a_v, linear_b_v``).

    Semantics implemented by the engine (reference image_encoder.py:42-48): ``qkv = W x + b``;
    ``qkv[..., :dim] += B_q A_q x``; ``qkv[..., -dim:] += B_v A_v x`` (no alpha/r scaling)."""

    def __init__(self, qkv: nn.Module, linear_a_q: nn.Module, linear_b_q: nn.Module, linear_a_v: nn.Module,
                 linear_b_v: nn.Module):
        super().__init__()
        self.qkv = qkv
        self.linear_a_q = linear_a_q
        self.linear_b_q = linear_b_q
        self.linear_a_v = linear_a_v
        self.linear_b_v = linear_b_v
        self.dim = qkv.in_features

    def forward(self, x):  # pragma: no cover - guard only
        raise RuntimeError("_LoRA_qkv_timm is evaluated inside the fused HIP QKV GEMM; call LoRA_ViT_timm instead")


class LoRA_ViT_timm(LoRAContainer):
    """``LoRA_ViT_timm(vit_model, r, num_classes, lora_layer)`` of the reference (image_encoder.py:51-109)."""

    def __init__(self, vit_model, r: int, num_classes: int = 0, lora_layer=None):
        super().__init__()
        # a falsy ``lora_layer`` (None or []) adapts every block (image_encoder.py:56-59)
        self._begin(vit_model, r, lora_layer if lora_layer else list(range(len(vit_model.blocks))))
        for index, block in enumerate(vit_model.blocks):
            if index in self.lora_layer:
                fused_qkv = block.attn.qkv
                self.dim = fused_qkv.in_features
                block.attn.qkv = _LoRA_qkv_timm(fused_qkv, *self._adapt(self.dim))
        self.reset_parameters()
        self.lora_vit = vit_model
        if num_classes > 0:
            self.reset_classifier(num_classes)

    def reset_classifier(self, num_classes):
        self.lora_vit.reset_classifier(num_classes=num_classes)
        self._engine = None

    def forward(self, x: Tensor) -> Tensor:
        from bioscanclip.hip.vit_engine import vit_forward
        return vit_forward(self, x)
