"""LoRA-adapted ViT-B/16 image encoder -- drop-in for reference ``bioscanclip/model/image_encoder.py``.

Same classes, constructor signatures, attribute names and ``state_dict`` keys as the reference
(``_LoRA_qkv_timm`` image_encoder.py:15-48, ``LoRA_ViT_timm`` :51-109), but the modules hold parameters
only: ``forward`` dispatches the whole encoder (patch-embed -> 12 pre-LN blocks with the rank-r LoRA
branch folded into the QKV GEMM's K dimension -> LN -> token 0 -> head) to the hand-written HIP engine
(``bioscanclip.hip``).  There is no torch fallback: on a machine without the built HIP library the call
raises.
"""
import math

import torch
import torch.nn as nn
from torch import Tensor


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


class LoRA_ViT_timm(nn.Module):
    def __init__(self, vit_model, r: int, num_classes: int = 0, lora_layer=None):
        super(LoRA_ViT_timm, self).__init__()

        assert r > 0
        self.r = r
        # reference image_encoder.py:56-59 -- NB ``if lora_layer:`` so ``[]`` means "all layers" 
        if lora_layer:
            self.lora_layer = lora_layer
        else:
            self.lora_layer = list(range(len(vit_model.blocks)))

        self.w_As = []  # plain lists, as in the reference (not ModuleList; App. A.5)
        self.w_Bs = []

        for param in vit_model.parameters():
            param.requires_grad = False

        for t_layer_i, blk in enumerate(vit_model.blocks):
            if t_layer_i not in self.lora_layer:
                continue
            w_qkv_linear = blk.attn.qkv
            self.dim = w_qkv_linear.in_features
            w_a_linear_q = nn.Linear(self.dim, r, bias=False)
            w_b_linear_q = nn.Linear(r, self.dim, bias=False)
            w_a_linear_v = nn.Linear(self.dim, r, bias=False)
            w_b_linear_v = nn.Linear(r, self.dim, bias=False)
            self.w_As.append(w_a_linear_q)
            self.w_Bs.append(w_b_linear_q)
            self.w_As.append(w_a_linear_v)
            self.w_Bs.append(w_b_linear_v)
            blk.attn.qkv = _LoRA_qkv_timm(w_qkv_linear, w_a_linear_q, w_b_linear_q, w_a_linear_v, w_b_linear_v)
        self.reset_parameters()
        self.lora_vit = vit_model
        if num_classes > 0:
            self.lora_vit.reset_classifier(num_classes=num_classes)
        self._engine = None

    def reset_classifier(self, num_classes):
        self.lora_vit.reset_classifier(num_classes=num_classes)
        self._engine = None

    def reset_parameters(self) -> None:
        for w_A in self.w_As:
            nn.init.kaiming_uniform_(w_A.weight, a=math.sqrt(5))
        for w_B in self.w_Bs:
            nn.init.zeros_(w_B.weight)

    def _load_from_state_dict(self, *args, **kwargs):
        self._engine = None  # frozen weights are re-packed to bf16 on the next forward
        return super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x: Tensor) -> Tensor:
        from bioscanclip.hip.vit_engine import vit_forward
        return vit_forward(self, x)
