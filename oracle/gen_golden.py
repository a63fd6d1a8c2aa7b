"""ORACLE pinning -- generates ``tests/golden/*.json`` by running the IMPORTED REFERENCE.

Runs only in the build container (needs ``/root/reference``); the GPU box never executes it.
    python oracle/gen_golden.py

What is run is the reference's own code: ``bioscanclip.model.loss_func.ContrastiveLoss``,
``bioscanclip.model.dna_encoder.LoRA_barcode_bert`` over HF ``BertForMaskedLM``,
``bioscanclip.model.language_encoder.LoRA_bert`` over HF ``BertModel``,
``bioscanclip.model.image_encoder.LoRA_ViT_timm`` over ``oracle/timm_like.VisionTransformer`` (timm is not
installed; see that file), ``bioscanclip.model.simple_clip.SimpleCLIP`` and a loop reproducing
``bioscanclip/epoch/train_epoch.py:22-44`` with ``torch.optim.AdamW`` (``train_cl.py:158``).  Packages the
reference imports but does not need on this path (torchtext, timm, open_clip, loratorch, clip, faiss, wandb,
torchvision, seaborn) are absent from the image and are stubbed in ``sys.modules``.

Weights and inputs come from ``oracle/synth.py`` (name-seeded), so fixtures hold expected outputs only.
Dropout is disabled in the HF configs (parity is defined for the deterministic path, SURVEY.md 7 "hard parts").
"""
import json
import os
import sys
from unittest.mock import MagicMock

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import transformers  # noqa: E402  (real; import before stubbing)
from transformers import BertConfig, BertForMaskedLM, BertModel  # noqa: E402

for name in ["torchtext", "torchtext.vocab", "timm", "timm.models", "timm.models.vision_transformer", "open_clip",
             "loratorch", "loratorch.layers", "clip", "faiss", "wandb", "torchvision", "torchvision.transforms",
             "seaborn", "h5py", "umap", "plotly", "plotly.express"]:
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = MagicMock()

sys.path.insert(0, "/root/reference")
from bioscanclip.model.loss_func import ContrastiveLoss  # noqa: E402  (reference)
from bioscanclip.model.dna_encoder import LoRA_barcode_bert  # noqa: E402  (reference)
from bioscanclip.model.language_encoder import LoRA_bert  # noqa: E402  (reference)
from bioscanclip.model.image_encoder import LoRA_ViT_timm  # noqa: E402  (reference)
from bioscanclip.model.simple_clip import SimpleCLIP  # noqa: E402  (reference)

from oracle import synth  # noqa: E402
from oracle.timm_like import VisionTransformer  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def summary(key, t):
    """Compact fingerprint of a tensor: shape, L2 norm, sum, first 8 values, dot with a name-seeded probe."""
    t = t.detach().to(torch.float64).reshape(-1)
    probe = synth.synth_tensor("probe." + key, t.shape, seed=7).to(torch.float64) / 0.02
    return {"shape": list(t.shape), "norm": t.norm().item(), "sum": t.sum().item(),
            "first": t[:8].tolist(), "probe": (t * probe).sum().item()}


def load_synth(module, seed, prefix=""):
    sd = synth.synth_state_dict({prefix + k: v for k, v in synth.shapes_of(module).items()}, seed)
    module.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    return sd


def build_dna(layers):
    cfg = BertConfig(vocab_size=1027, output_hidden_states=True, num_hidden_layers=layers,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    return LoRA_barcode_bert(BertForMaskedLM(cfg), r=4, num_classes=768)


def build_txt(layers):
    cfg = BertConfig(vocab_size=30522, hidden_size=512, num_hidden_layers=layers, num_attention_heads=8,
                     intermediate_size=2048, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = BertModel(cfg)
    for p in m.parameters():
        p.requires_grad = False
    return LoRA_bert(m, r=4, num_classes=768)


def build_vit(depth):
    return LoRA_ViT_timm(VisionTransformer(depth=depth), r=4, num_classes=768)


def grads_summary(module, prefix):
    return {prefix + k: summary(prefix + k, p.grad) for k, p in module.named_parameters()
            if p.requires_grad and p.grad is not None}


def gen_loss():
    out = {}
    crit = ContrastiveLoss(criterion=torch.nn.CrossEntropyLoss(), logit_scale=1 / 0.07)
    for N in (8, 64):
        for nmod in (2, 3):
            for dup in (False, True):
                name = f"N{N}_m{nmod}_{'dup' if dup else 'id'}"
                feats = [synth.synth_tensor(f"loss.{name}.{i}", (N, 768), seed=3).requires_grad_(True)
                         for i in range(nmod)]
                if dup:
                    label = torch.tensor([0, 0, 1, 2, 3, 3, 3, 4] * (N // 8))
                else:
                    label = torch.arange(N)
                img, dna = feats[0], feats[1]
                txt = feats[2] if nmod == 3 else None
                loss = crit(img, dna, txt, label)
                loss.backward()
                out[name] = {"loss": loss.item(), "label": label.tolist(),
                             "grads": [summary(f"loss.{name}.{i}", f.grad) for i, f in enumerate(feats)]}
    return out


def gen_encoders():
    out = {}
    # DNA: 2-layer and full-depth, B=2
    for layers in (2, 12):
        m = build_dna(layers)
        pref = "dna_encoder."
        load_synth(m, seed=11, prefix=pref)
        _, dna, _, _ = synth.synth_batch(2, seed=21)
        m.train()
        y = m(dna)
        w = synth.synth_tensor(f"dna.cot.{layers}", y.shape, seed=5)
        (y * w).sum().backward()
        out[f"dna_L{layers}"] = {"out": summary(f"dna.out.{layers}", y), "out_full_first_row": y[0, :16].tolist(),
                                 "grads": grads_summary(m, pref)}
    # text: 4 layers, B=4 with padding mask
    m = build_txt(4)
    pref = "language_encoder."
    load_synth(m, seed=12, prefix=pref)
    _, _, text, _ = synth.synth_batch(4, seed=22, with_text=True)
    m.train()
    y = m(text)
    w = synth.synth_tensor("txt.cot", y.shape, seed=5)
    (y * w).sum().backward()
    out["txt_L4"] = {"out": summary("txt.out", y), "out_full_first_row": y[0, :16].tolist(),
                     "grads": grads_summary(m, pref)}
    # ViT: depth 2 and 12, B=2
    for depth in (2, 12):
        m = build_vit(depth)
        pref = "image_encoder."
        load_synth(m, seed=13, prefix=pref)
        image, _, _, _ = synth.synth_batch(2, seed=23)
        m.train()
        y = m(image)
        w = synth.synth_tensor(f"vit.cot.{depth}", y.shape, seed=5)
        (y * w).sum().backward()
        out[f"vit_L{depth}"] = {"out": summary(f"vit.out.{depth}", y), "out_full_first_row": y[0, :16].tolist(),
                                "grads": grads_summary(m, pref)}
    return out


def gen_state_dict_keys():
    """Key names + shapes of the reference's full-size modules (App. A.5) for the drop-in check."""
    model = SimpleCLIP(build_vit(12), build_dna(12), build_txt(4))
    sd = model.state_dict()
    trainable = sorted(k for k, p in model.named_parameters() if p.requires_grad)
    return {"keys": {k: list(v.shape) for k, v in sd.items()}, "trainable": trainable,
            "n_trainable": sum(p.numel() for p in model.parameters() if p.requires_grad),
            "n_total": sum(p.numel() for p in model.parameters())}


def gen_trajectory(steps=10, B=8, with_text=False, seed=31, n_batches=2):
    """BASELINE config 1: Image+DNA two-tower, B=8, reference modules, AdamW lr 1e-3, 10 steps
    (loop body = train_epoch.py:22-44).  The loader cycles over ``n_batches`` distinct batches (a two-batch epoch repeated),
    so the loss falls from above ln B to a few percent of it within the run: a trajectory only a correct encoder +
    loss + backward + AdamW chain reproduces (with fresh random batches every step it would hover around ln B)."""
    torch.manual_seed(0)
    model = SimpleCLIP(build_vit(12), build_dna(12), build_txt(4) if with_text else None)
    load_synth(model, seed=seed)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=0.001)
    crit = ContrastiveLoss(criterion=torch.nn.CrossEntropyLoss(), logit_scale=1 / 0.07)
    losses = []
    first = {}
    for s in range(steps):
        image, dna, text, label = synth.synth_batch(B, seed=100 + s % n_batches, with_text=with_text)
        opt.zero_grad()
        img_o, dna_o, txt_o = model(image, dna, text)
        loss = crit(img_o, dna_o, txt_o, label)
        loss.backward()
        if s == 0:
            first = {"image_out": summary("traj.img", img_o), "dna_out": summary("traj.dna", dna_o),
                     "grads": {k: summary(k, p.grad) for k, p in model.named_parameters()
                               if p.requires_grad and p.grad is not None}}
            if txt_o is not None:
                first["text_out"] = summary("traj.txt", txt_o)
        opt.step()
        losses.append(loss.item())
        print(f"  step {s}: loss {loss.item():.6f}", flush=True)
    params = {k: summary(k, p) for k, p in model.named_parameters() if p.requires_grad}
    return {"losses": losses, "first_step": first, "params_after": params, "B": B, "steps": steps, "lr": 0.001,
            "weight_seed": seed, "batch_seed0": 100, "n_batches": n_batches}


def gen_retrieval():
    """The reference's own top_k_micro_accuracy / top_k_macro_accuracy (scripts/inference_and_eval.py:448-511) and
    convert_label_dict_to_list_of_dict (eval_epoch.py:26-38) on seeded inputs.  make_prediction itself needs faiss, which is
    not installed, so the search half is not pinned here (see oracle/retrieval.py)."""
    import importlib.util
    for name in ["hydra", "omegaconf", "matplotlib", "matplotlib.pyplot", "PIL", "sklearn.metrics"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = MagicMock()
    spec = importlib.util.spec_from_file_location("ref_inference_and_eval", "/root/reference/scripts/inference_and_eval.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from bioscanclip.epoch.eval_epoch import convert_label_dict_to_list_of_dict  # reference
    from oracle.retrieval import retrieval_case
    keys_label, gt_list, indices, pred_list = retrieval_case()
    k_list = [1, 3, 5]
    micro = ref.top_k_micro_accuracy(pred_list, gt_list, k_list=k_list)
    macro, per_class = ref.top_k_macro_accuracy(pred_list, gt_list, k_list=k_list)
    batch = {lv: [d[lv] for d in gt_list[:6]] for lv in ["order", "family", "genus", "species"]}
    return {"case": {"seed": 5, "n_keys": 60, "n_query": 40, "max_k": 5}, "k_list": k_list,
            "micro": {str(k): v for k, v in micro.items()}, "macro": {str(k): v for k, v in macro.items()},
            "per_class": {str(k): v for k, v in per_class.items()},
            "label_batch": batch, "label_list": convert_label_dict_to_list_of_dict(batch)}


def gen_fullft():
    """Full fine-tuning (disable_lora: true, simple_clip.py:151-201): the reference wrappers built with lora_layer=[] -- no LoRA
    in the BERT encoders, LoRA on every block of the ViT (an empty list is falsy in image_encoder.py:56-59) -- and EVERY
    parameter unfrozen; gradients of all of them, two layers each."""
    out = {}
    def unfreeze(m):
        for p in m.parameters():
            p.requires_grad = True
        return m
    def grads(m, pref):
        return {pref + k: summary(pref + k, p.grad) for k, p in m.named_parameters() if p.grad is not None}
    cfg = BertConfig(vocab_size=1027, output_hidden_states=True, num_hidden_layers=2, hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0)
    m = unfreeze(LoRA_barcode_bert(BertForMaskedLM(cfg), r=4, num_classes=768, lora_layer=[]))
    load_synth(m, seed=11, prefix="dna_encoder.")
    _, dna, _, _ = synth.synth_batch(2, seed=21)
    y = m.train()(dna)
    (y * synth.synth_tensor("dna.cot.ft", y.shape, seed=5)).sum().backward()
    out["dna"] = {"out": summary("dna.out.ft", y), "grads": grads(m, "dna_encoder.")}
    cfg = BertConfig(vocab_size=30522, hidden_size=512, num_hidden_layers=2, num_attention_heads=8, intermediate_size=2048,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = unfreeze(LoRA_bert(BertModel(cfg), r=4, num_classes=768, lora_layer=[]))
    load_synth(m, seed=12, prefix="language_encoder.")
    _, _, text, _ = synth.synth_batch(4, seed=22, with_text=True)
    y = m.train()(text)
    (y * synth.synth_tensor("txt.cot.ft", y.shape, seed=5)).sum().backward()
    out["txt"] = {"out": summary("txt.out.ft", y), "grads": grads(m, "language_encoder.")}
    m = unfreeze(LoRA_ViT_timm(VisionTransformer(depth=2), r=4, num_classes=768, lora_layer=[]))
    load_synth(m, seed=13, prefix="image_encoder.")
    image, _, _, _ = synth.synth_batch(2, seed=23)
    y = m.train()(image)
    (y * synth.synth_tensor("vit.cot.ft", y.shape, seed=5)).sum().backward()
    out["vit"] = {"out": summary("vit.out.ft", y), "grads": grads(m, "image_encoder.")}
    return out


def gen_pipeline():
    """PadSequence / KmerTokenizer of the reference (bioscanclip/util/util.py:48-69) on seeded strings: the pad + k-mer half
    of get_sequence_pipeline.  (Its vocab is torchtext's, absent here: the id map stays restated, see oracle/pipeline.py.)"""
    from bioscanclip.util.util import KmerTokenizer, PadSequence   # reference
    g = torch.Generator().manual_seed(9)
    seqs = []
    for i in range(12):
        L = int(torch.randint(0, 800, (1,), generator=g))
        s = "".join("ACGT"[int(c)] for c in torch.randint(0, 4, (L,), generator=g))
        if i % 3 == 1 and L > 40:
            s = s[:17] + "N" + s[18:30] + "-" + s[31:]
        seqs.append(s)
    seqs += ["", "ACGTA", "T" * 700]
    pad, tok = PadSequence(660), KmerTokenizer(5, stride=5)
    return {"seqs": seqs, "kmers": [tok(pad(s)) for s in seqs]}


def main():
    torch.set_num_threads(8)
    os.makedirs(GOLD, exist_ok=True)
    meta = {"torch": torch.__version__, "transformers": transformers.__version__,
            "reference": "bioscan-ml/bioscan-clip @ 2024-10-24 (/root/reference)"}
    jobs = {"loss": gen_loss, "encoders": gen_encoders, "state_dict_keys": gen_state_dict_keys,
            "trajectory_id": gen_trajectory, "retrieval": gen_retrieval, "pipeline": gen_pipeline, "fullft": gen_fullft,
            "trajectory_idt": lambda: gen_trajectory(steps=6, B=4, with_text=True, seed=32)}
    only = sys.argv[1:]
    for name, fn in jobs.items():
        if only and name not in only:
            continue
        print("generating", name, flush=True)
        data = fn()
        data["_meta"] = meta
        with open(os.path.join(GOLD, name + ".json"), "w") as f:
            json.dump(data, f, indent=1)


if __name__ == "__main__":
    main()
