"""How far apart are two equally valid bf16-operand evaluations of the same network?  (CPU only)

The bf16-rounding-aware oracle is evaluated twice: once accumulating in f32 (as the HIP kernels do, in another order) and
once accumulating in f64.  Both round every GEMM operand, q/k/v, P and the LoRA t to bf16 at the same points; they differ
only in the last bits of the accumulators, i.e. in which way a value that sits near a bf16 rounding boundary is rounded.
The distance between the two is the resolution of "parity with the emulating oracle": no implementation can be expected
to agree with either of them more closely than they agree with each other.
    python tools/emu_sensitivity.py [depth]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bioscan-clip_amd"), os.path.join(ROOT, "tests")]
from bioscanclip.model import arch  # noqa: E402
from bioscanclip.model.dna_encoder import LoRA_barcode_bert  # noqa: E402
from bioscanclip.model.image_encoder import LoRA_ViT_timm  # noqa: E402
from helpers import rel_err  # noqa: E402
from oracle import refcpu, synth  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 12
torch.set_num_threads(8)
m = LoRA_ViT_timm(arch.VisionTransformerParams(depth=depth), r=4, num_classes=768)
sd = synth.synth_state_dict({"image_encoder." + k: v for k, v in synth.shapes_of(m).items()}, 13)
image, dna, _, _ = synth.synth_batch(2, seed=23)
with torch.no_grad():
    t32, t64 = {}, {}
    y32 = refcpu.vit_encoder(sd, image, emulate_bf16=True, taps=t32)
    y64 = refcpu.vit_encoder({k: v.double() for k, v in sd.items()}, image.double(), emulate_bf16=True, taps=t64)
    yf = refcpu.vit_encoder({k: v.double() for k, v in sd.items()}, image.double())
print("ViT depth", depth)
for k in ["x0"] + [f"x{i}" for i in range(1, 2 * depth + 1)]:
    print(f"  {k:5s} emu(f32 acc) vs emu(f64 acc): {rel_err(t32[k], t64[k]):.2e}")
print(f"  out   emu32 vs emu64 {rel_err(y32, y64):.2e}   emu64 vs exact f64 {rel_err(y64, yf):.2e}")
NODROP = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
d = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=depth, **NODROP)), r=4, num_classes=768)
sdd = synth.synth_state_dict({"dna_encoder." + k: v for k, v in synth.shapes_of(d).items()}, 11)
with torch.no_grad():
    y32 = refcpu.barcode_bert_encoder(sdd, dna, emulate_bf16=True)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sdd.items()}
    y64 = refcpu.barcode_bert_encoder(sd64, dna, emulate_bf16=True)
    yf = refcpu.barcode_bert_encoder(sd64, dna)
print(f"BarcodeBERT depth {depth}: out emu32 vs emu64 {rel_err(y32, y64):.2e}   emu64 vs exact f64 {rel_err(y64, yf):.2e}")
