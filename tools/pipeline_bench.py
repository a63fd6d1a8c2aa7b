"""SURVEY 8f-3 measurement: the two per-sample input transforms on the GPU (inputs resident in HBM) with the CPU oracle beside
them.  Batch of B decoded uint8 images (BIOSCAN-like 341 x 256 crops, sizes jittered) -> f32 [B, 3, 224, 224];  B barcodes of
~660 nt -> int64 [B, 133].

    python tools/pipeline_bench.py [B]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))

from bioscanclip.hip import ops  # noqa: E402
from bioscanclip.util import gpu_pipeline as gp  # noqa: E402


def gpu_time(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    g = torch.Generator().manual_seed(0)
    images = []
    for _ in range(B):
        h = 256 + int(torch.randint(0, 120, (1,), generator=g))
        w = 256 + int(torch.randint(0, 120, (1,), generator=g))
        images.append((torch.rand(h, w, 3, generator=g) * 255).to(torch.uint8))
    aug = gp.GpuAugment(for_training=True, seed=1)
    out, params = aug(images)                     # builds the records; below the kernel alone is re-timed on resident buffers
    sizes = [(int(im.shape[0]), int(im.shape[1])) for im in images]
    resized = [gp._resized_size(h, w, 256) for h, w in sizes]
    src = torch.cat([im.reshape(-1).cuda() for im in images])
    cap = max(h1 * w1 for h1, w1 in resized)
    mid = torch.empty(B * 3 * cap, dtype=torch.float32, device="cuda")

    # re-create the record tensor exactly as GpuAugment does (its packing is private to __call__): run once through a hook
    rec_holder = {}
    orig = ops.augment_images

    def hook(src_, rec_, *a):
        rec_holder["rec"] = rec_
        return orig(src_, rec_, *a)
    ops.augment_images = hook
    gp.ops.augment_images = hook
    aug2 = gp.GpuAugment(for_training=True, seed=1)
    aug2(images, params=params)
    ops.augment_images = orig
    gp.ops.augment_images = orig
    rec = rec_holder["rec"]
    us = gpu_time(lambda: ops.augment_images(src, rec, B, cap, mid, 224, out))
    in_bytes = src.numel()
    mid_bytes = sum(h1 * w1 for h1, w1 in resized) * 3 * 4
    out_bytes = out.numel() * 4
    moved = in_bytes + 2 * mid_bytes + out_bytes          # uint8 in, resized f32 written + read back, f32 out
    print(f"augment_images  B={B}: {us:8.1f} us/batch = {B / us * 1e6:10.0f} images/s; {moved / 1e6:.1f} MB moved -> "
          f"{moved / us / 1e6:.2f} TB/s  (uint8 in {in_bytes / 1e6:.1f} MB, resized f32 {mid_bytes / 1e6:.1f} MB, out {out_bytes / 1e6:.1f} MB)")

    seqs = ["".join("ACGT"[int(c)] for c in torch.randint(0, 4, (640 + int(torch.randint(0, 40, (1,), generator=g)),), generator=g))
            for _ in range(B)]
    raw = [s.encode() for s in seqs]
    offsets = [0]
    for r in raw:
        offsets.append(offsets[-1] + len(r))
    blob = torch.frombuffer(bytearray(b"".join(raw)), dtype=torch.uint8).cuda()
    off = torch.tensor(offsets, dtype=torch.int64, device="cuda")
    ids = torch.empty(B, 133, dtype=torch.int64, device="cuda")
    us_t = gpu_time(lambda: ops.kmer_tokenize(blob, off, B, 660, 5, ids))
    print(f"kmer_tokenize   B={B}: {us_t:8.1f} us/batch = {B / us_t * 1e6:10.0f} barcodes/s ({blob.numel() / 1e3:.0f} KB in, {ids.numel() * 8 / 1e3:.0f} KB out)")

    # CPU oracle beside it (bounded sample)
    from oracle import pipeline as P
    n = min(B, 32)
    t0 = time.perf_counter()
    for im, p in zip(images[:n], params[:n]):
        P.augment(im, p)
    t_img = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    P.kmer_tokenize(seqs)
    t_tok = (time.perf_counter() - t0) / B
    print(f"CPU oracle ({torch.get_num_threads()} threads): augment {t_img * 1e3:.2f} ms/image = {1 / t_img:.0f} images/s; "
          f"tokenize {t_tok * 1e6:.1f} us/barcode = {1 / t_tok:.0f} barcodes/s")


if __name__ == "__main__":
    main()
