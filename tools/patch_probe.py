"""Patch-filter weight gradient of the full fine-tuning ViT engine vs dyp^T cols computed in f32 from the engine's own buffers, and
vs the 8-sample batch, at several batch sizes (found the shared transposed-operand buffer bug)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch
from oracle import synth
from bioscanclip.model import arch
from bioscanclip.model.image_encoder import LoRA_ViT_timm
from bioscanclip.hip.engine import split_plan

def run(B, n=8):
    torch.manual_seed(0)
    m = LoRA_ViT_timm(arch.VisionTransformerParams(depth=2), r=4, num_classes=768, lora_layer=[])
    sd = synth.synth_state_dict({"image_encoder." + k: v for k, v in synth.shapes_of(m).items()}, seed=13)
    m.load_state_dict({k[len("image_encoder."):]: v for k, v in sd.items()})
    for p in m.parameters(): p.requires_grad = True
    m.hip_full_ft = True
    m.to("cuda").train()
    image = synth.synth_batch(n, seed=71)[0]
    fill = synth.synth_batch(56, seed=72)[0]
    x = torch.cat([image, fill.repeat((B - n + 55) // 56, 1, 1, 1)[:B - n]]).cuda() if B > n else image.cuda()
    y = m(x)
    cot = synth.synth_tensor("c", (n, 768), seed=5).cuda()
    (y[:n] * cot).sum().backward()
    torch.cuda.synchronize()
    eng = m._engine
    ws = eng.ws
    M = B * 196
    ref = ws["dyp"][:M].float().t() @ ws["cols"][:M].float()
    g = m.lora_vit.patch_embed.proj.weight.grad.reshape(768, -1).clone()
    gb = m.lora_vit.patch_embed.proj.bias.grad.clone()
    refb = ws["dyp"][:M].float().sum(0)
    print(f"B={B} plan={split_plan(M, 768, 768)} grad vs dyp^T cols: {((g - ref).norm() / ref.norm()).item():.2e}  bias {((gb - refb).norm() / refb.norm()).item():.2e}  |g|={g.norm().item():.4e} |dyp|={ws['dyp'][:M].float().norm().item():.4e}")
    return g, ws["dyp"][:8*196].float().clone()

g8, d8 = run(8)
for B in (16, 64, 256):
    g, d = run(B)
    print(f"   vs B=8: grad {((g - g8).norm() / g8.norm()).item():.2e}   dyp(first 8 samples) {((d - d8).norm() / d8.norm()).item():.2e}")
