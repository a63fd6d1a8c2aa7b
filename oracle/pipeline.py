"""ORACLE -- test infrastructure only.  CPU restatement of the two input-pipeline transforms of SURVEY 8f-3.

* ``kmer_tokenize``: ``get_sequence_pipeline`` (reference bioscanclip/model/dna_encoder.py:25-35) = ``PadSequence(660)`` +
  ``KmerTokenizer(k, stride=k)`` (bioscanclip/util/util.py:48-69) + vocab lookup + the literal 0 in front.  The pad / k-mer
  half is PINNED on the reference's own classes (tests/golden/pipeline.json, produced by oracle/gen_golden.py importing
  them); the id map restates torchtext 0.15.2 ``build_vocab_from_iterator`` (absent here: equal-frequency tokens keep the
  iterator's order after the specials) -- that half is "parity unpinned".
* ``augment``: the training chain of ``Dataset_for_CL`` (bioscanclip/util/dataset.py:171-181): ToTensor, Resize(256,
  antialias), RandomResizedCrop(224, antialias), RandomHorizontalFlip, RandomVerticalFlip, RandomRotation((-45, 45)).
  torchvision (0.15.2 in the reference's requirements) is absent from this image; its tensor code path delegates the
  arithmetic to torch core ops that ARE here and are called below exactly as torchvision's functional_tensor does:
  ``F.interpolate(mode="bilinear", antialias=True, align_corners=False)`` (resize), slicing (crop), ``flip``, and
  ``F.grid_sample(mode="nearest", padding_mode="zeros", align_corners=False)`` over ``_gen_affine_grid`` of
  ``_get_inverse_affine_matrix`` (rotate).  The glue (output-size rule, matrix, grid) is restated from torchvision's
  published source: "parity unpinned" for that glue.
"""
import math
from itertools import product

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------------ tokeniser
def kmer_vocab(k=5):
    vocab = {"<MASK>": 0, "<CLS>": 1, "<UNK>": 2}
    for i, kmer in enumerate(product("ACGT", repeat=k)):
        vocab["".join(kmer)] = 3 + i
    return vocab


def pad_and_kmers(seq, k=5, max_len=660):
    """PadSequence(max_len) then KmerTokenizer(k, stride=k) (util.py:48-69)."""
    seq = seq[:max_len] if len(seq) > max_len else seq + "N" * (max_len - len(seq))
    return [seq[i:i + k] for i in range(0, len(seq) - k + 1, k)]


def kmer_tokenize(seqs, k=5, max_len=660):
    vocab = kmer_vocab(k)
    return torch.tensor([[0, *[vocab.get(t, 2) for t in pad_and_kmers(s, k, max_len)]] for s in seqs], dtype=torch.int64)


# ------------------------------------------------------------------------------------------------------ augmentation
def resized_size(h, w, size=256):
    """torchvision ``_compute_resized_output_size`` for an int size: the shorter side becomes ``size``."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def sample_params(h1, w1, generator, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), degrees=(-45.0, 45.0)):
    """``RandomResizedCrop.get_params`` + the two flips (p = 0.5) + ``RandomRotation.get_params``, drawn from ``generator``
    (the reference draws from torch's global generator; the draws are the caller's in the GPU version)."""
    u = lambda: torch.rand(1, generator=generator).item()
    area = h1 * w1
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    box = None
    for _ in range(10):
        target = area * (scale[0] + (scale[1] - scale[0]) * u())
        ar = math.exp(log_ratio[0] + (log_ratio[1] - log_ratio[0]) * u())
        w, h = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
        if 0 < w <= w1 and 0 < h <= h1:
            i = int(torch.randint(0, h1 - h + 1, (1,), generator=generator).item())
            j = int(torch.randint(0, w1 - w + 1, (1,), generator=generator).item())
            box = (i, j, h, w)
            break
    if box is None:  # fallback: central crop at the nearest valid ratio
        in_ratio = w1 / h1
        if in_ratio < ratio[0]:
            w, h = w1, int(round(w1 / ratio[0]))
        elif in_ratio > ratio[1]:
            h, w = h1, int(round(h1 * ratio[1]))
        else:
            w, h = w1, h1
        box = ((h1 - h) // 2, (w1 - w) // 2, h, w)
    hflip, vflip = u() < 0.5, u() < 0.5
    angle = degrees[0] + (degrees[1] - degrees[0]) * u()
    return {"box": box, "hflip": hflip, "vflip": vflip, "angle": angle}


def eval_params(h1, w1, size=224):
    """Resize(256) -> CenterCrop(224) (dataset.py:183-200): a centred box, no flips, no rotation."""
    top, left = int(round((h1 - size) / 2.0)), int(round((w1 - size) / 2.0))
    return {"box": (top, left, size, size), "hflip": False, "vflip": False, "angle": 0.0}


def _resize(img, size_hw):
    return F.interpolate(img[None], size=list(size_hw), mode="bilinear", align_corners=False, antialias=True)[0]


def _rotate_nearest(img, angle):
    """torchvision functional_tensor.rotate(img, matrix=_get_inverse_affine_matrix([0,0], -angle, [0,0], 1, [0,0]))."""
    rot = math.radians(-angle)
    matrix = [math.cos(rot), math.sin(rot), 0.0, -math.sin(rot), math.cos(rot), 0.0]
    _, h, w = img.shape
    theta = torch.tensor(matrix, dtype=img.dtype).reshape(1, 2, 3)
    d = 0.5
    base = torch.empty(1, h, w, 3, dtype=img.dtype)
    base[..., 0].copy_(torch.linspace(-w * 0.5 + d, w * 0.5 + d - 1, steps=w))
    base[..., 1].copy_(torch.linspace(-h * 0.5 + d, h * 0.5 + d - 1, steps=h).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=img.dtype)
    grid = base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2)
    return F.grid_sample(img[None], grid, mode="nearest", padding_mode="zeros", align_corners=False)[0]


def augment(img_u8_hwc, params, out_size=224, resize_to=256):
    """One image through the chain with the given draws.  img: uint8 [H, W, 3]."""
    x = img_u8_hwc.permute(2, 0, 1).to(torch.float32).div(255)              # ToTensor
    h1, w1 = resized_size(x.shape[1], x.shape[2], resize_to)
    x = _resize(x, (h1, w1))                                               # Resize(256, antialias=True)
    i, j, h, w = params["box"]
    x = _resize(x[:, i:i + h, j:j + w], (out_size, out_size))               # RandomResizedCrop / CenterCrop
    if params["hflip"]:
        x = x.flip(-1)
    if params["vflip"]:
        x = x.flip(-2)
    if params["angle"] != 0.0:
        x = _rotate_nearest(x, params["angle"])                            # RandomRotation: nearest, zeros outside
    return x
