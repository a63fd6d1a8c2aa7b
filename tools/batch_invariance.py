"""Debug: do rows 0..7 of a B-sample forward equal the 8-sample forward?  Prints where (which sub-layer) they part."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bioscan-clip_amd"), os.path.join(ROOT, "tests")]
from helpers import rel_err
from oracle import synth
from bioscanclip.model import arch
from bioscanclip.model.image_encoder import LoRA_ViT_timm
from bioscanclip.model.dna_encoder import LoRA_barcode_bert
NODROP = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
which = sys.argv[1] if len(sys.argv) > 1 else "vit"
Bs = [int(x) for x in sys.argv[2:]] or [8, 16, 64, 256]
if which == "vit":
    m = LoRA_ViT_timm(arch.vit_base_patch16_224(), r=4, num_classes=768)
    pre = "image_encoder."
else:
    m = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(**NODROP)), r=4, num_classes=768)
    pre = "dna_encoder."
sd = synth.synth_state_dict({pre + k: v for k, v in synth.shapes_of(m).items()}, 61)
m.load_state_dict({k[len(pre):]: v for k, v in sd.items()})
m.cuda().eval()
image, dna, _, _ = synth.synth_batch(8, seed=71)
fi, fd, _, _ = synth.synth_batch(56, seed=72)
base = None
for B in Bs:
    reps = (B - 8 + 55) // 56 if B > 8 else 0
    if which == "vit":
        x = torch.cat([image, fi.repeat(reps, 1, 1, 1)[:B - 8]]) if B > 8 else image
    else:
        x = torch.cat([dna, fd.repeat(reps, 1)[:B - 8]]) if B > 8 else dna
    with torch.no_grad():
        y = m(x.cuda())[:8].clone()
        y2 = m(x.cuda())[:8].clone()
    ws = m._engine.ws
    S = 197 if which == "vit" else 133
    keys = ["x"] if which == "vit" else ["s1", "s2"]
    taps = {}
    if which == "vit":
        for i, t in enumerate(ws["x"]):
            taps[f"x{i}"] = t[:8 * S].clone()
        for i, t in enumerate(ws["qkv"]):
            taps[f"qkv{i}"] = t[:8 * S].float().clone()
        for i, t in enumerate(ws["ctx"]):
            taps[f"ctx{i}"] = t[:8 * S].float().clone()
        for i, t in enumerate(ws["h1"]):
            taps[f"h1_{i}"] = t[:8 * S].float().clone()
    else:
        for i, t in enumerate(ws["s1"]):
            taps[f"s1_{i}"] = t[:8 * S].clone()
        for i, t in enumerate(ws["s2"]):
            taps[f"s2_{i}"] = t[:8 * S].clone()
    print(f"B={B}: rerun bitwise equal: {torch.equal(y, y2)}")
    if base is None:
        base = (y, taps)
    else:
        print(f"   out vs B=8: {rel_err(y, base[0]):.3e}")
        for k in taps:
            if k.startswith("x") and which == "vit" and int(k[1:]) >= 23:
                continue  # last block: token-0 rows only
            e = rel_err(taps[k], base[1][k])
            if e > 0:
                print(f"   first difference at {k}: {e:.3e}")
                break
