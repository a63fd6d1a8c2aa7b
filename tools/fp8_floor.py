"""What does e4m3 cost, and would activation scales help?  CPU emulation of the fp8 trunks (oracle/refcpu.py emulate_fp8; ViT, depth 6):
e4m3 activations with scale 1 (what the engines do) / per-tensor amax / per-token amax, e4m3 weights with per-row scales, and each
operand class alone; every variant against the f32 evaluation.      python tools/fp8_floor.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
from oracle import refcpu, synth  # noqa: E402
from bioscanclip.model import arch  # noqa: E402
from bioscanclip.model.image_encoder import LoRA_ViT_timm  # noqa: E402

torch.set_num_threads(8)
FP8 = torch.float8_e4m3fn
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
orig = refcpu._q8
MODE = {"act": "one", "w": True}


def q8(x, scale=None):
    if scale is not None:                       # a weight (per-row scale)
        return orig(x, scale) if MODE["w"] else x
    if MODE["act"] == "none":
        return x
    if MODE["act"] == "one":
        return orig(x)
    s = x.detach().abs().max() / 448.0 if MODE["act"] == "tensor" else (x.detach().abs().amax(dim=-1, keepdim=True) / 448.0).clamp_min(1e-30)
    return orig(x, s)


refcpu._q8 = q8
depth = 6
m = LoRA_ViT_timm(arch.VisionTransformerParams(depth=depth), r=4, num_classes=768)
sd = synth.synth_state_dict({"image_encoder." + k: v for k, v in synth.shapes_of(m).items()}, 7)
image, _, _, _ = synth.synth_batch(2, seed=4)
with torch.no_grad():
    y0 = refcpu.vit_encoder(sd, image)
    yb = refcpu.vit_encoder(sd, image, emulate_bf16=True)
    print(f"ViT depth {depth}, B = 2.  bf16 emulation vs f32: {rel(yb, y0):.3e}")
    for act, w, label in (("one", True, "e4m3 weights (per-row scale) + e4m3 activations, scale 1  [the engines]"),
                          ("tensor", True, "... activations with a per-tensor amax scale"),
                          ("row", True, "... activations with a per-token amax scale"),
                          ("none", True, "e4m3 weights only"), ("one", False, "e4m3 activations only (scale 1)")):
        MODE["act"], MODE["w"] = act, w
        y = refcpu.vit_encoder(sd, image, emulate_fp8=True)
        print(f"  {label:85s} vs f32 {rel(y, y0):.3e}   vs bf16 emulation {rel(y, yb):.3e}")
