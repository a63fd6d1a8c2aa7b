"""Writes tests/golden/tiny_shard: 24 seeded samples in the shard layout of bioscanclip/util/shards.py (test fixture: data only --
random pixels, random nucleotide strings, random token ids; nothing of the reference).   python oracle/gen_tiny_shard.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
from bioscanclip.util import shards  # noqa: E402

n = 24
rng = np.random.default_rng(20261004)
yy, xx = np.mgrid[0:96, 0:96]
images = []
for i in range(n):
    h, w = int(rng.integers(48, 72)), int(rng.integers(56, 90))
    # smooth structure + noise, so that the antialiased resampling is exercised on something other than white noise
    base = 127 + 90 * np.sin(0.07 * (i + 3) * yy[:h, :w] + 0.11 * xx[:h, :w])[..., None] * np.array([1.0, 0.6, -0.8])
    images.append(np.clip(base + rng.normal(0, 20, (h, w, 3)), 0, 255).astype(np.uint8))
barcodes = []
for i in range(n):
    L = int(rng.integers(560, 720))                      # shorter and longer than the tokeniser's 660
    s = rng.choice(list("ACGT"), size=L)
    if i % 5 == 0:
        s[rng.integers(0, L, 3)] = "N"                   # ambiguous bases -> <UNK> k-mers
    barcodes.append("".join(s))
lens = rng.integers(6, 21, n)
mask = (np.arange(20)[None] < lens[:, None]).astype(np.int64)
ids = rng.integers(1000, 30522, (n, 20)) * mask
ids[:, 0] = 101
ids[np.arange(n), lens - 1] = 102
taxa = {"order": [f"Order{i % 2}" for i in range(n)], "family": [f"Family{i % 3}" for i in range(n)],
        "genus": [f"Genus{i % 4}" for i in range(n)], "species": [f"Genus{i % 4} sp{i % 6}" for i in range(n)]}
shards.write_shard(os.path.join(ROOT, "tests", "golden", "tiny_shard"), images, barcodes, ids, np.zeros((n, 20), np.int64), mask,
                   [f"TINY{i:04d}" for i in range(n)], taxonomy=taxa, split="train", dataset="bioscan_1m")
print("wrote", n, "samples")
