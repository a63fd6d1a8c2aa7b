"""Which operand roundings carry the default path's distance to the f32 reference?  (CPU only, zero GPU minutes)

The rounding-aware oracle (oracle/refcpu.py, emulate_bf16=True) rounds every operand where the HIP kernels round it.  This tool
re-evaluates it with one rounding SITE (or a group of sites) left exact at a time -- refcpu.EXACT_SITES -- and prints the
distance of the embedding and of the worst trainable-gradient tensor to the plain f32 evaluation: the sites whose removal moves
the distance are the ones a mixed-precision mode (hip/engine.py BSCLIP_PARITY=3) has to run on split operands.
    python tools/site_sensitivity.py [depth] [vit|dna|both]
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
which = sys.argv[2] if len(sys.argv) > 2 else "both"
torch.set_num_threads(8)

GROUPS = [
    ("(default: every site rounded)", []),
    ("resid", ["resid"]),
    ("qkv.a", ["qkv.a"]), ("qkv.w", ["qkv.w"]), ("qkv GEMM (a + w)", ["qkv.a", "qkv.w"]),
    ("q + k (score operands)", ["q", "k"]), ("v", ["v"]), ("p", ["p"]),
    ("attention operands q, k, v, p", ["q", "k", "v", "p"]),
    ("qkv GEMM + q, k", ["qkv.a", "qkv.w", "q", "k"]),
    ("qkv GEMM + q, k, v, p", ["qkv.a", "qkv.w", "q", "k", "v", "p"]),
    ("proj GEMM", ["proj.a", "proj.w"]),
    ("fc1 GEMM", ["fc1.a", "fc1.w"]), ("fc2 GEMM", ["fc2.a", "fc2.w"]), ("MLP (fc1 + fc2)", ["fc1.a", "fc1.w", "fc2.a", "fc2.w"]),
    ("head", ["head.a", "head.w"]), ("lora", ["lora"]),
    ("attention half: qkv GEMM, q k v p, proj", ["qkv.a", "qkv.w", "q", "k", "v", "p", "proj.a", "proj.w"]),
    ("attention half + resid", ["qkv.a", "qkv.w", "q", "k", "v", "p", "proj.a", "proj.w", "resid"]),
    ("all activations (a sides, q k v p, resid)", ["qkv.a", "q", "k", "v", "p", "proj.a", "fc1.a", "fc2.a", "resid", "head.a", "lora"]),
    ("all weights (w sides)", ["qkv.w", "proj.w", "fc1.w", "fc2.w", "head.w"]),
    ("everything but the MLP", ["qkv.a", "qkv.w", "q", "k", "v", "p", "proj.a", "proj.w", "resid", "head.a", "head.w", "lora"]),
    ("everything", ["qkv.a", "qkv.w", "q", "k", "v", "p", "proj.a", "proj.w", "fc1.a", "fc1.w", "fc2.a", "fc2.w", "resid", "head.a",
                    "head.w", "lora"]),
]


def run(name, encode, sd, x):
    keys = [k for k in sd if refcpu.is_trainable_key(k)]

    def evaluate(emulate):
        leaf = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
        y = encode(leaf, x, emulate)
        g = torch.Generator().manual_seed(5)
        cot = torch.randn(y.shape, generator=g)
        (y * cot).sum().backward()
        return y.detach(), {k: leaf[k].grad for k in keys}

    refcpu.EXACT_SITES = set()
    y_ref, g_ref = evaluate(False)
    print(f"{name} depth {depth}: distance to the f32 evaluation with the named sites left EXACT")
    for label, sites in GROUPS:
        refcpu.EXACT_SITES = set(sites)
        y, g = evaluate(True)
        worst = max(rel_err(g[k], g_ref[k]) for k in keys if g_ref[k] is not None and float(g_ref[k].norm()) > 0)
        print(f"  {label:48s} embedding {rel_err(y, y_ref):.2e}   worst gradient {worst:.2e}", flush=True)
    refcpu.EXACT_SITES = set()


if which in ("vit", "both"):
    m = LoRA_ViT_timm(arch.VisionTransformerParams(depth=depth), r=4, num_classes=768)
    sd = synth.synth_state_dict({"image_encoder." + k: v for k, v in synth.shapes_of(m).items()}, 13)
    image, dna, _, _ = synth.synth_batch(2, seed=23)
    run("ViT", lambda s, x, e: refcpu.vit_encoder(s, x, emulate_bf16=e), sd, image)
if which in ("dna", "both"):
    NODROP = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    d = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=depth, **NODROP)), r=4, num_classes=768)
    sdd = synth.synth_state_dict({"dna_encoder." + k: v for k, v in synth.shapes_of(d).items()}, 11)
    image, dna, _, _ = synth.synth_batch(2, seed=23)
    run("BarcodeBERT", lambda s, x, e: refcpu.barcode_bert_encoder(s, x, emulate_bf16=e), sdd, dna)
