"""SURVEY 8f-3 on the GPU: the 5-mer tokeniser and the image augmentation chain against the CPU oracle (oracle/pipeline.py)
on seeded inputs and seeded draws.  Tokens are bit-exact.  Images: the resampling is f32 arithmetic in the same order as
torch's (1e-6); the nearest-neighbour rotation picks a source pixel by rounding a coordinate, so a handful of pixels that sit
within float error of a half-integer may pick the neighbour -- they are counted and bounded, everything else is compared."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pipeline as P  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _seqs(n, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        L = int(torch.randint(0, 900, (1,), generator=g))            # shorter and longer than 660, and empty
        s = "".join("ACGT"[int(c)] for c in torch.randint(0, 4, (L,), generator=g))
        if i % 3 == 1 and L > 40:                                     # ambiguity codes, gaps, lower case -> <UNK>
            s = s[:17] + "N" + s[18:30] + "-" + s[31:37] + "a" + s[38:]
        out.append(s)
    return out + ["", "ACGTA", "ACGT", "T" * 660, "G" * 2000]


def test_kmer_tokenizer_bit_exact():
    from bioscanclip.util.gpu_pipeline import tokenize_barcodes
    seqs = _seqs(300, 3)
    want = P.kmer_tokenize(seqs)
    got = tokenize_barcodes(seqs).cpu()
    assert got.dtype == torch.int64 and tuple(got.shape) == (len(seqs), 133)
    assert torch.equal(got, want)
    assert (got[:, 0] == 0).all() and int(got.max()) <= 1026 and (got[-5, 1:] == 2).all()     # "" -> all <UNK>
    # product drop-in: identical to the reference-named CPU pipeline kept in bioscanclip.model.dna_encoder
    from bioscanclip.model.dna_encoder import get_sequence_pipeline
    pipe = get_sequence_pipeline(5)
    assert got[:40].tolist() == [pipe(s) for s in seqs[:40]]


def _images(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(256, 256), (300, 280), (197, 341), (512, 384), (64, 80), (256, 1024)]
    ims = []
    for h, w in shapes:
        base = torch.rand(h // 8 + 2, w // 8 + 2, 3, generator=g)          # smooth content + noise: both filters matter
        up = torch.nn.functional.interpolate(base.permute(2, 0, 1)[None], size=(h, w), mode="bilinear")[0].permute(1, 2, 0)
        ims.append(((0.7 * up + 0.3 * torch.rand(h, w, 3, generator=g)) * 255).to(torch.uint8))
    return ims


@pytest.mark.parametrize("for_training", [True, False])
def test_augmentation_matches_oracle(for_training):
    from bioscanclip.util.gpu_pipeline import GpuAugment
    ims = _images(11)
    aug = GpuAugment(for_training=for_training, seed=5)
    out, params = aug(ims)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (len(ims), 3, 224, 224)
    # the draws are the reference's distributions: the oracle's sampler with the same generator state gives the same draws
    g = torch.Generator().manual_seed(5)
    mism_total = 0
    for b, im in enumerate(ims):
        h1, w1 = P.resized_size(im.shape[0], im.shape[1])
        want_p = P.sample_params(h1, w1, g) if for_training else P.eval_params(h1, w1)
        assert want_p == params[b]
        want = P.augment(im, want_p)
        got = out[b].cpu()
        diff = (got - want).abs()
        bad = diff.amax(0) > 2e-5                                          # pixels (all channels) that disagree
        mism_total += int(bad.sum())
        assert bad.float().mean().item() < 2e-3, (b, want_p, bad.float().mean().item())
        assert diff[:, ~bad].max().item() <= 2e-5
        if want_p["angle"] != 0.0:
            assert (want == 0).all(0).float().mean() > 0.01               # the rotation left black corners: it really ran
    if not for_training:
        assert mism_total == 0                                             # no rotation -> no rounding ambiguity at all


def test_train_cl_on_raw_synthetic_data(tmp_path, capsys):
    """scripts/train_cl.py with dataset=synthetic_raw: every batch goes uint8 images / nucleotide strings -> GPU augmentation +
    tokeniser -> encoders -> loss -> AdamW."""
    import os
    import sys
    scripts = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bioscan-clip_amd", "scripts")
    sys.path.insert(0, scripts)
    import train_cl
    losses = train_cl.main(["model_config=lora_vit_lora_barcode_bert_ssl", "model_config.batch_size=8", "model_config.epochs=1",
                            "synthetic_steps_per_epoch=3", "dataset=synthetic_raw", f"project_root_path={tmp_path}"])
    assert len(losses) == 1 and losses[0] == losses[0] and 0.5 < losses[0] < 10
