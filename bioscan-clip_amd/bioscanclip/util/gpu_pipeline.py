"""Input pipeline on the GPU: the reference's per-sample CPU transforms as two HIP kernels.

* ``tokenize_barcodes(seqs)``: ``get_sequence_pipeline(k=5)`` (reference bioscanclip/model/dna_encoder.py:25-35) for a batch of
  nucleotide strings -> int64 [B, 133] on the GPU (``bsclip_kmer_tokenize``).
* ``GpuAugment``: the training / evaluation image transforms of ``Dataset_for_CL`` (reference bioscanclip/util/dataset.py:
  171-200) for a batch of DECODED uint8 images (H x W x 3, sizes may differ): ToTensor, Resize(256, antialias),
  RandomResizedCrop(224, antialias) | CenterCrop(224), random flips, RandomRotation(+-45 deg) -> f32 [B, 3, 224, 224]
  (``bsclip_augment_images``).  The random draws are made here on the host from a seeded ``torch.Generator`` (a few scalars
  per image), the arithmetic runs on the device.  JPEG decoding and the HDF5 reader stay outside.
"""
import math
import struct

import torch

from bioscanclip.hip import ops


def tokenize_barcodes(seqs, k=5, max_len=660, device="cuda"):
    """list of str / bytes -> int64 [B, max_len // k + 1] on ``device``: id 0 first, then the non-overlapping k-mer ids."""
    raw = [s.encode("ascii", "replace") if isinstance(s, str) else bytes(s) for s in seqs]
    offsets = [0]
    for r in raw:
        offsets.append(offsets[-1] + len(r))
    blob = torch.frombuffer(bytearray(b"".join(raw) or b"\0"), dtype=torch.uint8).to(device)
    off = torch.tensor(offsets, dtype=torch.int64, device=device)
    ids = torch.empty(len(raw), max_len // k + 1, dtype=torch.int64, device=device)
    ops.kmer_tokenize(blob, off, len(raw), max_len, k, ids)
    return ids


def _resized_size(h, w, size):
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


class GpuAugment:
    """``for_training=True``: the reference's training chain; ``False``: Resize(256) -> CenterCrop(224)."""

    def __init__(self, for_training=True, out_size=224, resize_to=256, seed=0, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0),
                 degrees=(-45.0, 45.0)):
        self.for_training, self.out_size, self.resize_to = for_training, out_size, resize_to
        self.scale, self.ratio, self.degrees = scale, ratio, degrees
        self.generator = torch.Generator().manual_seed(seed)

    # -- the random draws: RandomResizedCrop.get_params, two flips at p = 0.5, RandomRotation.get_params -------------------
    def sample(self, h1, w1):
        if not self.for_training:
            s = self.out_size
            return {"box": (int(round((h1 - s) / 2.0)), int(round((w1 - s) / 2.0)), s, s), "hflip": False, "vflip": False,
                    "angle": 0.0}
        g = self.generator
        u = lambda: torch.rand(1, generator=g).item()
        area = h1 * w1
        lo, hi = math.log(self.ratio[0]), math.log(self.ratio[1])
        box = None
        for _ in range(10):
            target = area * (self.scale[0] + (self.scale[1] - self.scale[0]) * u())
            ar = math.exp(lo + (hi - lo) * u())
            w, h = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
            if 0 < w <= w1 and 0 < h <= h1:
                i = int(torch.randint(0, h1 - h + 1, (1,), generator=g).item())
                j = int(torch.randint(0, w1 - w + 1, (1,), generator=g).item())
                box = (i, j, h, w)
                break
        if box is None:
            in_ratio = w1 / h1
            if in_ratio < self.ratio[0]:
                w, h = w1, int(round(w1 / self.ratio[0]))
            elif in_ratio > self.ratio[1]:
                h, w = h1, int(round(h1 * self.ratio[1]))
            else:
                w, h = w1, h1
            box = ((h1 - h) // 2, (w1 - w) // 2, h, w)
        hflip, vflip = u() < 0.5, u() < 0.5
        return {"box": box, "hflip": hflip, "vflip": vflip,
                "angle": self.degrees[0] + (self.degrees[1] - self.degrees[0]) * u()}

    def _records(self, sizes, resized, params):
        """The 16-int parameter record per image that ``bsclip_augment_images`` reads (csrc/pipeline.hip)."""
        f2i = lambda f: struct.unpack("<i", struct.pack("<f", f))[0]
        rec, off = [], 0
        for (h0, w0), (h1, w1), p in zip(sizes, resized, params):
            i, j, h, w = p["box"]
            if not (0 <= i and 0 <= j and 0 < h and 0 < w and i + h <= h1 and j + w <= w1):
                raise ValueError(f"crop box {p['box']} outside the {h1}x{w1} resized image")
            rot = math.radians(-p["angle"])        # torchvision: _get_inverse_affine_matrix(centre 0, -angle), f64 then f32
            lo = off & 0xFFFFFFFF
            rec += [lo - (1 << 32) if lo >= (1 << 31) else lo, off >> 32, h0, w0, h1, w1,
                    i, j, h, w, int(p["hflip"]), int(p["vflip"]), int(p["angle"] != 0.0), f2i(math.cos(rot)), f2i(-math.sin(rot)), 0]
            off += h0 * w0 * 3
        return rec

    def run_packed(self, src, sizes, params=None, device="cuda"):
        """``src``: uint8 device buffer holding the images back to back (H x W x 3 each, ``sizes`` = [(H, W), ...]) -- the form the
        shard loader stages.  Returns (f32 [B, 3, S, S], the draws used)."""
        B = len(sizes)
        resized = [_resized_size(h, w, self.resize_to) for h, w in sizes]
        if params is None:
            params = [self.sample(h1, w1) for h1, w1 in resized]
        rec = self._records(sizes, resized, params)
        cap = max(h1 * w1 for h1, w1 in resized)
        mid = torch.empty(B * 3 * cap, dtype=torch.float32, device=device)
        out = torch.empty(B, 3, self.out_size, self.out_size, dtype=torch.float32, device=device)
        ops.augment_images(src, torch.tensor(rec, dtype=torch.int32).to(device, non_blocking=True), B, cap, mid, self.out_size, out)
        return out, params

    def __call__(self, images, params=None, device="cuda"):
        """images: list of uint8 tensors [H, W, 3] (CPU or GPU).  Returns (f32 [B, 3, S, S] on ``device``, the draws used)."""
        sizes = [(int(im.shape[0]), int(im.shape[1])) for im in images]
        src = torch.cat([im.reshape(-1).to(device=device, dtype=torch.uint8) for im in images])
        return self.run_packed(src, sizes, params, device=device)
