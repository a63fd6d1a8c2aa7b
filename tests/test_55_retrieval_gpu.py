"""Retrieval row: `bsclip_topk_ip` through the C ABI and the inference / evaluation mirrors against the
numpy oracle (oracle/retrieval.py).

Bar: key INDICES bit-exact and in the oracle's order, except where the oracle's own float64 scores of the two candidates
differ by less than 2e-6 (the split-bf16 GEMM is f32-accurate, not f64); similarities within 2e-6 absolute.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import refcpu, synth  # noqa: E402
from oracle import retrieval as R  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd", "scripts"))
TOL_SIM = 2e-6


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _check(q, keys, k):
    from bioscanclip.hip import ops
    sims, idx = ops.topk_ip(torch.from_numpy(q).cuda(), torch.from_numpy(keys).cuda(), k)
    sims, idx = sims.cpu().numpy(), idx.cpu().numpy()
    ref_s, ref_i = R.topk_ip(q, keys, k)
    assert sims.shape == ref_s.shape and idx.dtype == np.int64
    assert np.abs(sims - ref_s).max() <= TOL_SIM
    assert (np.diff(sims, axis=1) <= 0).all(), "similarities must be sorted in decreasing order"
    bad = np.argwhere(idx != ref_i)
    if len(bad):
        qn, kn = R.l2_normalize_rows(q).astype(np.float64), R.l2_normalize_rows(keys).astype(np.float64)
        for r, c in bad:  # only near-ties of the f64 scores may be ordered differently
            a, b = qn[r] @ kn[idx[r, c]], qn[r] @ kn[ref_i[r, c]]
            assert abs(a - b) <= TOL_SIM, (r, c, idx[r], ref_i[r], a, b)
    return len(bad)


@pytest.mark.parametrize("Q,K,D,k", [(300, 5000, 768, 5), (7, 1000, 768, 1), (1300, 2500, 768, 3), (64, 777, 1536, 16),
                                     (1, 5, 64, 5)])
def test_topk_matches_oracle(Q, K, D, k):
    rng = np.random.RandomState(Q + K)
    q = rng.randn(Q, D).astype(np.float32) * 3.0  # not unit norm: the entry point normalises, like make_prediction
    keys = rng.randn(K, D).astype(np.float32) * 0.2
    keys[: min(Q, K) // 2] = q[: min(Q, K) // 2] * 0.5 + 0.3 * keys[: min(Q, K) // 2]  # a planted neighbour per query
    _check(q, keys, k)


def test_topk_ties_zero_rows_and_self_retrieval():
    from bioscanclip.hip import ops
    rng = np.random.RandomState(3)
    base = rng.randn(200, 768).astype(np.float32)
    keys = np.concatenate([base, base, np.zeros((3, 768), np.float32)])  # exact duplicates + all-zero keys
    sims, idx = ops.topk_ip(torch.from_numpy(base).cuda(), torch.from_numpy(keys).cuda(), 2)
    idx = idx.cpu().numpy()
    assert (idx[:, 0] == np.arange(200)).all() and (idx[:, 1] == np.arange(200) + 200).all(), "ties -> lower index first"
    assert np.abs(sims.cpu().numpy() - 1.0).max() <= TOL_SIM
    _check(np.concatenate([base[:5], np.zeros((1, 768), np.float32)]), keys, 5)  # a zero query scores 0 everywhere


def test_topk_many_equal_scores_takes_exact_fallback():
    """All keys identical: every score ties, the candidate buffer overflows and the full-insertion path must still return
    the k lowest indices in order; then a row where 300 keys tie for first place ahead of distinct ones."""
    from bioscanclip.hip import ops
    rng = np.random.RandomState(4)
    one = rng.randn(1, 768).astype(np.float32)
    q = rng.randn(9, 768).astype(np.float32)
    keys = np.repeat(one, 700, axis=0)
    _, idx = ops.topk_ip(torch.from_numpy(q).cuda(), torch.from_numpy(keys).cuda(), 16)
    assert (idx.cpu().numpy() == np.arange(16)[None]).all()
    keys2 = rng.randn(900, 768).astype(np.float32)
    keys2[5:900:3] = q[0] * 2.0  # 299 keys parallel to query 0
    _check(q, keys2, 5)
    _, idx2 = ops.topk_ip(torch.from_numpy(q[:1].copy()).cuda(), torch.from_numpy(keys2).cuda(), 5)
    assert idx2.cpu().numpy().tolist() == [[5, 8, 11, 14, 17]]


def test_topk_full_size_self_retrieval():
    """BIOSCAN-1M key-set size (21 118 keys): size-independent property -- every key retrieves itself first with
    similarity 1, and the scores are sorted."""
    from bioscanclip.hip import ops
    g = torch.Generator(device="cuda").manual_seed(0)
    keys = torch.randn(21118, 768, device="cuda", generator=g)
    sims, idx = ops.topk_ip(keys[:4096].contiguous(), keys, 5)
    assert (idx[:, 0] == torch.arange(4096, device="cuda")).all()
    assert (sims[:, 0] - 1).abs().max().item() <= TOL_SIM
    assert (sims[:, 1:] <= sims[:, :-1]).all()


def test_topk_rejects_bad_arguments():
    from bioscanclip.hip import ops
    q = torch.randn(4, 768, device="cuda")
    with pytest.raises(ValueError):
        ops.topk_ip(q, torch.randn(3, 768, device="cuda"), 5)  # k > K
    with pytest.raises(ValueError):
        ops.topk_ip(q, torch.randn(30, 768, device="cuda"), 17)
    with pytest.raises(ValueError):
        ops.topk_ip(q.cpu(), torch.randn(30, 768), 5)  # no CPU path


def test_make_prediction_and_accuracy_match_oracle():
    import inference_and_eval as host
    keys_label, gt_list, _, _ = R.retrieval_case(seed=9, n_keys=500, n_query=120)
    rng = np.random.RandomState(1)
    keys = rng.randn(500, 768)
    query = keys[rng.randint(0, 500, 120)] + 0.9 * rng.randn(120, 768)
    pred, sims, idx = host.make_prediction(query, keys, keys_label, with_similarity=True, with_indices=True, max_k=5)
    ref = R.make_prediction(query, keys, keys_label, max_k=5)
    assert pred == ref
    assert sims.shape == (120, 5) and idx.shape == (120, 5)
    assert host.top_k_micro_accuracy(pred, gt_list, k_list=[1, 3, 5]) == R.top_k_micro_accuracy(ref, gt_list, [1, 3, 5])
    assert host.top_k_macro_accuracy(pred, gt_list) == R.top_k_macro_accuracy(ref, gt_list, [1, 3, 5])


class _EvalLoader:
    """Batches in the layout of the reference's evaluation loaders: 7-tuple with a dict of taxonomy-name lists."""

    def __init__(self, n_batches, B, seed):
        self.batches = []
        for s in range(n_batches):
            image, dna, text, _ = synth.synth_batch(B, seed=seed + s, with_text=True)
            lab = {lv: [f"{lv[0]}{(s * B + i) % m}" for i in range(B)] for lv, m in
                   zip(["order", "family", "genus", "species"], [2, 3, 5, 7])}
            self.batches.append(([f"P{s}_{i}" for i in range(B)], image, dna, text["input_ids"], text["token_type_ids"],
                                 text["attention_mask"], lab))

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


def test_get_feature_and_label_matches_oracle():
    """inference_epoch.get_feature_and_label on a 2-layer I+D+T model vs the CPU oracle's normalised encoder outputs."""
    from bioscanclip.epoch.inference_epoch import get_feature_and_label
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    from bioscanclip.model.language_encoder import LoRA_bert
    from bioscanclip.model.simple_clip import SimpleCLIP
    import inference_and_eval as host
    model = SimpleCLIP(LoRA_ViT_timm(arch.VisionTransformerParams(depth=2), r=4, num_classes=768),
                       LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=2)), r=4,
                                         num_classes=768),
                       LoRA_bert(arch.BertModelParams(arch.bert_small_config()), r=4, num_classes=768))
    sd = synth.synth_state_dict(synth.shapes_of(model), 41)
    model.load_state_dict(sd)
    model.to("cuda").train()  # get_feature_and_label must switch to eval (dropout 0.1 is configured)
    loader = _EvalLoader(2, 3, seed=50)
    with pytest.raises(TypeError):
        get_feature_and_label(loader, model, "cuda", type_of_feature="audio")
    split = host.get_features_and_label(loader, model, "cuda", for_key_set=True)
    assert not model.training
    assert split["file_name_list"] == [f"P{s}_{i}" for s in range(2) for i in range(3)]
    assert split["label_list"][4] == {"order": "o0", "family": "f1", "genus": "g4", "species": "s4"}
    image = torch.cat([b[1] for b in loader.batches])
    dna = torch.cat([b[2] for b in loader.batches])
    text = {k: torch.cat([b[i] for b in loader.batches]) for k, i in
            [("input_ids", 3), ("token_type_ids", 4), ("attention_mask", 5)]}
    with torch.no_grad():
        ref = {"encoded_image_feature": refcpu.l2_normalize(refcpu.vit_encoder(sd, image)),
               "encoded_dna_feature": refcpu.l2_normalize(refcpu.barcode_bert_encoder(sd, dna)),
               "encoded_language_feature": refcpu.l2_normalize(refcpu.bert_text_encoder(sd, text))}
    for key, r in ref.items():
        got = split[key]
        assert got.dtype == np.float64 and got.shape == (6, 768)
        assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-5
        err = np.linalg.norm(got - r.numpy()) / np.linalg.norm(r.numpy())
        assert err <= 2e-2, (key, err)  # TOL_EMB_F32 of test_20_encoders_gpu.py
    assert split["averaged_feature"].shape == (6, 768) and split["concatenated_feature"].shape == (6, 1536)
    assert split["all_key_features"].shape == (18, 768) and len(split["all_key_features_label"]) == 18
    # a model without a text tower returns (None, None, None) for it, as the reference does
    model.language_encoder = None
    assert get_feature_and_label(loader, model, "cuda", type_of_feature="text") == (None, None, None)
    # end to end: seen/unseen queries against the key set
    acc, per_class, pred = host.inference_and_print_result(split, split, split, k_list=[1, 3, 5])
    cell = acc["encoded_image_feature"]["encoded_image_feature"]["seen"]
    assert cell["micro_acc"][1]["species"] == 1.0 and cell["macro_acc"][5]["order"] == 1.0  # self retrieval
    assert acc["concatenated_feature"]["encoded_image_feature"] == {}  # 1536-d queries skip 768-d keys (:669-676)
    assert acc["concatenated_feature"]["concatenated_feature"]["unseen"]["micro_acc"][1]["genus"] == 1.0
