"""Host-side logic that needs no GPU: drop-in surface (state_dict keys, constructor semantics, error behaviour),
config loader, k-mer pipeline, flat-parameter plumbing."""
import os

import pytest
import torch

from helpers import load_golden

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bioscan-clip_amd")


def _build_clip():
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    from bioscanclip.model.language_encoder import LoRA_bert
    from bioscanclip.model.simple_clip import SimpleCLIP
    return SimpleCLIP(LoRA_ViT_timm(arch.vit_base_patch16_224(), r=4, num_classes=768),
                      LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config()), r=4, num_classes=768),
                      LoRA_bert(arch.BertModelParams(arch.bert_small_config()), r=4, num_classes=768))


def test_state_dict_keys_and_trainable_set_match_reference():
    """Keys/shapes captured from the imported reference (oracle/gen_golden.py:gen_state_dict_keys)."""
    g = load_golden("state_dict_keys")
    model = _build_clip()
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == g["keys"]
    assert sorted(k for k, p in model.named_parameters() if p.requires_grad) == g["trainable"]
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == g["n_trainable"] == 1902848
    assert sum(p.numel() for p in model.parameters()) == g["n_total"]


def test_lora_init_and_lora_layer_semantics():
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    m = LoRA_ViT_timm(arch.VisionTransformerParams(depth=3), r=4, num_classes=768)
    assert all((w.weight == 0).all() for w in m.w_Bs) and all(w.weight.abs().sum() > 0 for w in m.w_As)
    assert len(m.w_As) == 6 and m.lora_layer == [0, 1, 2]
    # reference quirk: ViT treats [] as "all layers", the BERT wrappers as "no layers"
    assert LoRA_ViT_timm(arch.VisionTransformerParams(depth=2), r=4, lora_layer=[]).lora_layer == [0, 1]
    d = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=2)), r=4, lora_layer=[])
    assert d.lora_layer == [] and len(d.w_As) == 0
    with pytest.raises(AssertionError):
        LoRA_ViT_timm(arch.VisionTransformerParams(depth=1), r=0)


def test_no_cpu_compute_path():
    """The product must fail loudly instead of computing on the CPU."""
    from bioscanclip.model import arch
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    from bioscanclip.model.loss_func import ContrastiveLoss
    m = LoRA_ViT_timm(arch.VisionTransformerParams(depth=1), r=4, num_classes=768)
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(1, 3, 224, 224))
    with pytest.raises(RuntimeError):
        m.lora_vit(torch.zeros(1, 3, 224, 224))
    crit = ContrastiveLoss(torch.nn.CrossEntropyLoss(), 1 / 0.07)
    with pytest.raises(ValueError, match="Too less element"):
        crit(torch.zeros(4, 768), None, None, torch.arange(4))
    with pytest.raises(RuntimeError, match="GPU"):
        crit(torch.zeros(4, 768), torch.zeros(4, 768), None, torch.arange(4))


def test_sequence_pipeline_matches_reference_semantics():
    """dna_encoder.py:25-35 + util.py:48-69: 660 nt -> [0] + 132 5-mer ids, 'N' k-mers -> <UNK>=2."""
    from bioscanclip.model.dna_encoder import get_sequence_pipeline, kmer_vocab
    pipe = get_sequence_pipeline(5)
    v = kmer_vocab(5)
    assert len(v) == 1027 and v["AAAAA"] == 3 and v["AAAAC"] == 4 and v["TTTTT"] == 1026
    ids = pipe("ACGTA" * 10)
    assert len(ids) == 133 and ids[0] == 0
    assert ids[1] == 3 + (0 * 256 + 1 * 64 + 2 * 16 + 3 * 4 + 0)
    assert ids[11:] == [2] * 122           # padding with 'N' -> <UNK>
    assert len(pipe("A" * 1000)) == 133    # truncation
    assert pipe("ACGTN" + "A" * 655)[1] == 2


def test_config_loader_hydra_like():
    from bioscanclip.util.config import load_config
    cfg = load_config(os.path.join(PKG, "bioscanclip", "config"),
                      ["model_config=lora_vit_lora_barcode_bert_ssl", "model_config.batch_size=8",
                       "model_config.lr_config.lr=0.0005", "model_config.lr_scheduler=cosine"])
    mc = cfg.model_config
    assert mc.batch_size == 8 and mc.image.input_type == "image" and mc.dna.model == "lora_barcode_bert"
    assert not hasattr(mc, "language") and hasattr(mc, "lr_config") and mc.lr_config.lr == 0.0005
    assert mc.output_dim == 768 and cfg.debug_flag is True
    with pytest.raises(FileNotFoundError):
        load_config(os.path.join(PKG, "bioscanclip", "config"), ["model_config=mlp_ssl"])


def test_load_clip_model_config_keys():
    from bioscanclip.model.simple_clip import load_clip_model
    from bioscanclip.util.config import load_config
    cfg = load_config(os.path.join(PKG, "bioscanclip", "config"), ["model_config=lora_vit_lora_barcode_bert_lora_bert_ssl"])
    model = load_clip_model(cfg)
    assert model.image_encoder is not None and model.dna_encoder is not None and model.language_encoder is not None
    n_lora_regime = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert n_lora_regime == 1_902_848                                   # I+D+T trainable count
    # disable_lora: true (full fine-tuning, reference simple_clip.py:151-201): lora_layer=[] everywhere -- no LoRA in the BERT
    # encoders, LoRA on every ViT block ([] is falsy in image_encoder.py:56-59) -- and every parameter unfrozen
    cfg.model_config.disable_lora = True
    ft = load_clip_model(cfg)
    assert all(p.requires_grad for p in ft.parameters())
    assert len(ft.dna_encoder.w_As) == 0 and len(ft.language_encoder.w_As) == 0 and len(ft.image_encoder.w_As) == 24
    assert all(getattr(e, "hip_full_ft", False) for e in (ft.image_encoder, ft.dna_encoder, ft.language_encoder))
    assert sum(p.numel() for p in ft.parameters()) > 200_000_000


def test_flat_params_alias_parameters_and_grads():
    from bioscanclip.hip.engine import FlatParams
    a, b = torch.nn.Parameter(torch.randn(3, 5)), torch.nn.Parameter(torch.randn(7))
    a0, b0 = a.detach().clone(), b.detach().clone()
    flat = FlatParams([a, b], torch.device("cpu"))
    assert torch.equal(a, a0) and torch.equal(b, b0) and flat.valid()
    flat.data.mul_(2)
    assert torch.equal(a, 2 * a0) and torch.equal(b, 2 * b0)
    flat.grad.fill_(3.0)
    assert (a.grad == 3).all() and (b.grad == 3).all()
    a.grad = None
    b.grad = None
    flat.bind_grads()          # dropped grads -> buffer cleared, aliases restored
    assert (flat.grad == 0).all() and a.grad.data_ptr() == flat.grad.data_ptr()
    a.data = a.data.clone()
    assert not flat.valid()


def test_synthetic_loader_layout():
    from bioscanclip.util.synthetic import SyntheticCLIPLoader
    batch = next(iter(SyntheticCLIPLoader(4, 2, with_text=True)))
    pid, image, dna, ids, tt, am, label = batch
    assert image.shape == (4, 3, 224, 224) and 0 <= image.min() and image.max() < 1
    assert dna.shape == (4, 133) and (dna[:, 0] == 0).all() and dna[:, 1:].min() >= 3 and dna.max() <= 1026
    assert ids.shape == (4, 20) and (ids[:, 0] == 101).all() and am.sum(1).min() >= 6 and len(pid) == 4


def test_checkpoint_helpers_and_roundtrip(tmp_path):
    """Reference util.py:72-84 semantics + a state_dict written by one model loads strictly into another (same keys)."""
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    from bioscanclip.util.util import load_bert_model, remove_extra_pre_fix
    assert remove_extra_pre_fix({"module.a.b": 1, "c": 2, "module.module.d": 3}) == {"a.b": 1, "c": 2, "module.d": 3}
    torch.manual_seed(0)
    bert = arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=1))
    path = str(tmp_path / "bert.pth")
    torch.save({"module." + k: v for k, v in bert.state_dict().items()}, path)
    torch.manual_seed(1)
    other = arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=1))
    load_bert_model(other, path)
    assert all(torch.equal(v, other.state_dict()[k]) for k, v in bert.state_dict().items())
    with pytest.raises(RuntimeError):
        load_bert_model(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=2)), path)  # strict
    a = LoRA_barcode_bert(bert, r=4, num_classes=768)
    b = LoRA_barcode_bert(other, r=4, num_classes=768)
    ck = str(tmp_path / "last.pth")
    torch.save(a.state_dict(), ck)
    b.load_state_dict(torch.load(ck))
    assert all(torch.equal(v, b.state_dict()[k]) for k, v in a.state_dict().items())


def test_split_plan_for_weight_gradient_gemms():
    """hip/engine.py:split_plan -- the K-slice plan of bsclip_gemm_splitk_f32 (full fine-tuning dW GEMMs): equal slices of whole
    64-wide K-tiles, about two rounds of the 256 CUs, no split where the output already fills the chip or the reduction is short."""
    from bioscanclip.hip.engine import split_plan
    for M, N, K in [(50432, 768, 3072), (50432, 3072, 768), (50432, 2304, 768), (50432, 768, 768), (34048, 768, 768),
                    (5120, 512, 2048), (5120, 1536, 512), (1064, 768, 768)]:
        S, Mp = split_plan(M, N, K)
        tiles = -(-N // 256) * (K // 256)
        assert S >= 1 and Mp >= M and Mp % (64 * S) == 0 and Mp - M < 64 * S
        assert S * tiles <= 512 and Mp // S >= 256                      # at most two rounds; >= four K-tiles per slice
        if M >= 16384:
            assert S * tiles > 256                                       # more than one round of the chip
    assert split_plan(394, 768, 3072) == (1, 448)                        # fixture-sized batches: one slice, padded to 64
    assert split_plan(50432, 768, 1000)[0] == 1                          # K not a multiple of the 256-wide tile: the plain kernel
    assert split_plan(50432, 4096, 4096)[0] == 1                         # 256 output tiles already fill the chip


def test_shipped_model_configs_carry_the_reference_values():
    """The on-path reference configurations (config/model_config/**; VERDICT r2 missing #5) by file name, with the values the path
    reads: batch size, epochs, towers, disable_lora, LR schedule and its bounds."""
    import os
    import yaml
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bioscan-clip_amd", "bioscanclip", "config",
                        "model_config")
    IDT, ID, IT = ("image", "dna", "language"), ("image", "dna"), ("image", "language")
    want = {
        "lora_vit_lora_barcode_bert_5m": (400, 4, ID, False, None, None),
        "lora_vit_lora_barcode_bert_lora_bert_5m": (400, 4, IDT, False, None, None),
        "lora_with_batch_size_300/lora_vit_lora_barcode_bert_lora_bert_ssl_cosin_lr_sche": (300, 15, IDT, False, "cosine", dict(lr=1e-3, min_lr=1e-5)),
        "lora_with_batch_size_300/lora_vit_lora_barcode_bert_lora_bert_ssl_one_cycle_lr_sche": (300, 15, IDT, False, "one_cycle", dict(lr=1e-5, max_lr=1e-3)),
    }
    for towers, name in ((ID, "image_dna"), (IDT, "image_dna_text"), (IT, "image_text")):
        want[f"full_fine_tuning/cosin/BIOSCAN_1M_{name}_cosin_lr_sche"] = (300, 15, towers, True, "cosine", dict(lr=5e-5, min_lr=1e-5))
        want[f"full_fine_tuning/one_cycle/BIOSCAN_1M_{name}_one_cycle_lr_sche"] = (300, 15, towers, True, "one_cycle", dict(lr=1e-6, max_lr=5e-5))
    for rel, (bs, ep, towers, ft, sched, lrc) in want.items():
        cfg = yaml.safe_load(open(os.path.join(root, rel + ".yaml")))
        assert cfg["batch_size"] == bs and cfg["epochs"] == ep, rel
        assert tuple(k for k in ("image", "dna", "language") if k in cfg) == towers, rel
        assert bool(cfg.get("disable_lora", False)) == ft and cfg.get("lr_scheduler") == sched, rel
        if lrc is not None:
            assert {k: float(v) for k, v in cfg["lr_config"].items()} == lrc, rel


def test_product_package_is_oracle_free():
    """The oracle is test infrastructure: nothing under the product package may import it (the smoke checker that does lives at
    the repo root, next to __graft_entry__.py)."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bioscan-clip_amd")
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)", re.M)
    hits = []
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py") and pat.search(open(os.path.join(d, f)).read()):
                hits.append(os.path.join(d, f))
    assert not hits, hits
