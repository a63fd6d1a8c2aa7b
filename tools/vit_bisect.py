"""Per-sub-layer bisection of the HIP ViT forward against the CPU oracle (f32 and bf16-rounding-aware).

Prints, for every tensor the engine keeps (residual stream after each sub-layer, LN1 output, qkv, ctx), the normwise
relative error vs the f32 oracle and vs the oracle that rounds where the HIP path rounds.  Run on the GPU box:
    python tools/vit_bisect.py [depth] [batch]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bioscan-clip_amd"), os.path.join(ROOT, "tests")]

from bioscanclip.model import arch  # noqa: E402
from bioscanclip.model.image_encoder import LoRA_ViT_timm  # noqa: E402
from helpers import rel_err  # noqa: E402
from oracle import refcpu, synth  # noqa: E402


def main():
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    m = LoRA_ViT_timm(arch.VisionTransformerParams(depth=depth), r=4, num_classes=768)
    sd = synth.synth_state_dict({"image_encoder." + k: v for k, v in synth.shapes_of(m).items()}, 13)
    m.load_state_dict({k[len("image_encoder."):]: v for k, v in sd.items()})
    image, _, _, _ = synth.synth_batch(B, seed=23)
    m.to("cuda").eval()
    with torch.no_grad():
        y = m(image.cuda())
    torch.cuda.synchronize()
    ws = m._engine.ws
    S, H = 197, 768
    taps_f, taps_e = {}, {}
    with torch.no_grad():
        yf = refcpu.vit_encoder(sd, image, taps=taps_f)
        ye = refcpu.vit_encoder(sd, image, emulate_bf16=True, taps=taps_e)
    rows = []

    bf = lambda t: t.to(torch.bfloat16).float()

    def cmp(name, got, key, tok0=False, col0=0):
        rounded = got.dtype == torch.bfloat16   # tensors the engine stores in bf16 are compared with the rounded taps
        got = got.float().cpu().reshape(B, S, -1)
        f, e = taps_f[key], taps_e[key]
        if rounded:
            e = bf(e)
        got = got[..., col0:col0 + f.shape[-1]]
        if tok0:
            got, f, e = got[:, :1], f[:, :1], e[:, :1]
        rows.append({"tensor": name, "vs_f32": rel_err(got, f), "vs_emu": rel_err(got, e), "emu_vs_f32": rel_err(e, f)})

    cmp("x0 (patch+cls+pos)", ws["x"][0], "x0")
    for l in range(depth):
        last = l == depth - 1
        cmp(f"h1.{l} (LN1 out)", ws["h1"][l], f"h1.{l}")
        cmp(f"t.{l} (LoRA y.A^T)", ws["h1"][l], f"t.{l}", col0=H)
        cmp(f"qkv.{l}", ws["qkv"][l], f"qkv.{l}")
        cmp(f"ctx.{l}", ws["ctx"][l], f"ctx.{l}", tok0=last)
        cmp(f"x{2 * l + 1} (after attn)", ws["x"][2 * l + 1], f"x{2 * l + 1}", tok0=last)
        cmp(f"x{2 * l + 2} (after mlp)", ws["x"][2 * l + 2], f"x{2 * l + 2}", tok0=last)
    rows.append({"tensor": "out", "vs_f32": rel_err(y, yf), "vs_emu": rel_err(y, ye), "emu_vs_f32": rel_err(ye, yf)})
    print(f"{'tensor':28s} {'hip vs f32':>11s} {'hip vs emu':>11s} {'emu vs f32':>11s}")
    for r in rows:
        print(f"{r['tensor']:28s} {r['vs_f32']:11.2e} {r['vs_emu']:11.2e} {r['emu_vs_f32']:11.2e}")
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/vit_bisect_d{depth}.json", "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
