"""Entry used by ``LoRA_ViT_timm.forward`` (reference image_encoder.py:108-109)."""
import torch

from .engine import ViTEngine, run_encoder, wants_fp8, wants_full_ft


def vit_forward(module, x):
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        raise RuntimeError("LoRA_ViT_timm.forward: the image batch must live on the GPU "
                           "(bioscanclip has no CPU compute path)")
    x = x.to(torch.float32).contiguous()
    def build():
        if wants_full_ft(module):   # disable_lora: true -- every parameter trained (hip/engine_ft.py)
            from .engine_ft import ViTEngineFT
            return ViTEngineFT(module, x.device)
        return ViTEngine(module, x.device, fp8=wants_fp8(module))
    return run_encoder(module, build, (x,))
