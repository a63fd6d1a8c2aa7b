"""Shared LoRA bookkeeping of the three encoder containers (image / DNA / text).

The containers only hold parameters under the reference's attribute names -- the arithmetic is in the HIP engines -- so what
they share is: freezing the wrapped trunk, creating the rank-r ``A`` / ``B`` pairs in the reference's order (``A_q, B_q, A_v,
B_v`` per adapted layer, so a seeded construction draws the same random numbers), the reference's initialisation (Kaiming
uniform ``A``, zero ``B``) and dropping the packed bf16 engine when a checkpoint is loaded."""
import math

import torch.nn as nn


def lora_pair(dim, rank):
    """``(A, B)`` = ``(Linear(dim -> rank), Linear(rank -> dim))``, both without bias."""
    return nn.Linear(dim, rank, bias=False), nn.Linear(rank, dim, bias=False)


class LoRAContainer(nn.Module):
    """Base of ``LoRA_ViT_timm`` / ``LoRA_barcode_bert`` / ``LoRA_bert``: ``r``, ``lora_layer``, ``w_As``, ``w_Bs`` (plain lists,
    like the reference's, so they do not show up in ``state_dict``) and the lazily built HIP engine."""

    def _begin(self, trunk, rank, adapted_layers):
        if rank <= 0:
            raise AssertionError("LoRA rank r must be positive")
        self.r = rank
        self.lora_layer = adapted_layers
        self.w_As, self.w_Bs = [], []
        trunk.requires_grad_(False)
        self._engine = None

    def _adapt(self, dim):
        """New (A_q, B_q, A_v, B_v) for one layer, registered in ``w_As`` / ``w_Bs`` in the reference's order."""
        a_q, b_q = lora_pair(dim, self.r)
        a_v, b_v = lora_pair(dim, self.r)
        self.w_As += [a_q, a_v]
        self.w_Bs += [b_q, b_v]
        return a_q, b_q, a_v, b_v

    def reset_parameters(self) -> None:
        for a in self.w_As:
            nn.init.kaiming_uniform_(a.weight, a=math.sqrt(5))
        for b in self.w_Bs:
            nn.init.zeros_(b.weight)

    def _load_from_state_dict(self, *args, **kwargs):
        self._engine = None  # frozen weights are re-packed to bf16 on the next forward
        return super()._load_from_state_dict(*args, **kwargs)
