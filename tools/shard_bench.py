"""SURVEY 8f-3, data side: does the shard loader keep up with the step?  Writes a synthetic shard of N BIOSCAN-sized decoded images
(256 x 341 +- jitter) to /tmp, then measures (1) the loader alone, (2) the I+D training step at local batch 256 fed by it (captured
hipGraph, inputs copied into the graph's static buffers) next to the same step on resident inputs.

    python tools/shard_bench.py [N=3072] [B=256]
"""
import os
import shutil
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import bench  # noqa: E402
from bioscanclip.util import shards  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    d = "/tmp/bsclip_shard_bench"
    shutil.rmtree(d, ignore_errors=True)
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()

    def images():
        base = rng.integers(0, 256, (400, 400, 3), dtype=np.uint8)
        for i in range(N):
            h, w = 256 + int(rng.integers(0, 32)), 341 + int(rng.integers(-40, 40))
            yield np.roll(base, i * 7, axis=1)[:h, :w]
    bc = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(600, 700)))) for _ in range(N)]
    ids = rng.integers(1000, 30522, (N, 20))
    shards.write_shard(d, images(), bc, ids, np.zeros((N, 20), np.int64), np.ones((N, 20), np.int64), [f"S{i}" for i in range(N)])
    nbytes = os.path.getsize(os.path.join(d, "images.bin"))
    print(f"shard: {N} images, {nbytes / 1e6:.0f} MB of decoded pixels, written in {time.perf_counter() - t0:.1f} s", flush=True)

    # (1) the loader alone (two epochs: the first warms the page cache and the staging buffers)
    ld = shards.ShardLoader(d, B, shuffle=True, seed=1, for_training=True, with_text=False)
    for epoch in range(3):
        ld.set_epoch(epoch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for pid, image, dna, *_ in ld:
            n += image.shape[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"loader alone, epoch {epoch}: {n} images in {dt * 1e3:.0f} ms = {n / dt:,.0f} images/s "
              f"({nbytes / N * n / dt / 1e9:.2f} GB/s of uint8 through pinned staging + H2D + augmentation + tokeniser)", flush=True)

    # (2) the step fed by the loader vs resident inputs
    from bioscanclip.hip.graph import GraphedStep
    from bioscanclip.hip.optim import FusedAdamW
    from bioscanclip.model.loss_func import ContrastiveLoss
    dev = torch.device("cuda", 0)
    model = bench.build_model(False, dev)
    model.train()
    opt = FusedAdamW(model.parameters(), lr=1e-3)
    g = GraphedStep(model, opt, ContrastiveLoss(torch.nn.CrossEntropyLoss(), 1 / 0.07), warmup=2)
    image, dna, _ = bench.synthetic_batch(B, False, dev, seed=1)
    label = torch.arange(B, device=dev)
    for _ in range(6):
        g(image, dna, None, label)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(12):
        g(image, dna, None, label)
    torch.cuda.synchronize()
    resident = (time.perf_counter() - t0) / 12 * 1e3
    for epoch in range(2):
        ld.set_epoch(10 + epoch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = steps = 0
        for pid, im, dn, _, _, _, lab in ld:
            if im.shape[0] != B:
                continue
            g(im, dn, None, lab)
            n += B
            steps += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"step fed by the loader, epoch {epoch}: {steps} steps, {dt / steps * 1e3:.2f} ms/step = {n / dt:,.0f} images/s "
              f"(resident inputs: {resident:.2f} ms/step = {B / resident * 1e3:,.0f} images/s)", flush=True)
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
