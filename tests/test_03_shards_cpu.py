"""The data side of the input pipeline (reference bioscanclip/util/dataset.py:41-48, 97-275) without a GPU: the shard
format round trip, the committed tiny shard, and the sample order -- torch's own DistributedSampler(drop_last=True), as the
reference's prepare() builds it."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
TINY = os.path.join(ROOT, "tests", "golden", "tiny_shard")


def test_shard_round_trip(tmp_path):
    from bioscanclip.util import shards
    rng = np.random.default_rng(1)
    n = 7
    imgs = [rng.integers(0, 256, (20 + i, 31 + 2 * i, 3), dtype=np.uint8) for i in range(n)]
    bc = ["".join(rng.choice(list("ACGTN"), size=100 + 13 * i)) for i in range(n)]
    ids, am = rng.integers(0, 30522, (n, 20)), rng.integers(0, 2, (n, 20))
    taxa = {t: [f"{t}-{i % 3}" for i in range(n)] for t in ("order", "family", "genus", "species")}
    shards.write_shard(str(tmp_path), iter(imgs), bc, ids, np.zeros((n, 20), int), am, [f"id{i}" for i in range(n)],
                       labels=np.arange(n)[::-1], taxonomy=taxa, split="val", dataset="bioscan_5m")
    assert shards.is_shard(str(tmp_path)) and not shards.is_shard(str(tmp_path / "nope"))
    sh = shards.Shard(str(tmp_path))
    assert len(sh) == n and sh.meta["split"] == "val" and sh.meta["dataset"] == "bioscan_5m" and sh.meta["text_len"] == 20
    for i in range(n):
        assert np.array_equal(sh.image(i), imgs[i]) and sh.barcode(i) == bc[i]
        assert sh.processid[i].decode() == f"id{i}" and int(sh.labels[i]) == n - 1 - i
    assert np.array_equal(sh.input_ids, ids) and np.array_equal(sh.attention_mask, am)
    assert sh.label_dicts([2, 5]) == [{t: f"{t}-{i % 3}" for t in taxa} for i in (2, 5)]
    (tmp_path / "meta.json").write_text('{"format": "something-else", "version": 1, "n": 7}')
    with pytest.raises(ValueError, match="not a bsclip-shard"):
        shards.Shard(str(tmp_path))


def test_committed_tiny_shard_is_the_generators_output():
    from bioscanclip.util import shards
    sh = shards.Shard(TINY)
    assert len(sh) == 24 and sh.image(0).dtype == np.uint8 and sh.image(0).shape[2] == 3
    assert all(set(sh.barcode(i)) <= set("ACGTN") for i in range(24))
    assert {len(sh.barcode(i)) > 660 for i in range(24)} == {True, False}     # both sides of the tokeniser's pad / truncate point
    assert sh.processid[5].decode() == "TINY0005" and np.array_equal(sh.labels, np.arange(24))


@pytest.mark.parametrize("n,world", [(24, 1), (24, 2), (25, 2), (23, 4), (1000, 8)])
def test_sample_order_is_the_reference_samplers(n, world):
    """prepare() (dataset.py:41-48): DistributedSampler(num_replicas, rank, shuffle, drop_last=True): equal counts, disjoint
    ranks, the tail dropped, one order per (seed, epoch), every rank drawing from the same permutation."""
    from bioscanclip.util.shards import rank_indices
    per = n // world if n % world == 0 else -(-(n - world) // world)      # torch's drop_last rule
    for shuffle in (False, True):
        got = [rank_indices(n, r, world, shuffle, seed=3, epoch=2) for r in range(world)]
        assert all(len(g) == per for g in got)
        flat = [i for g in got for i in g]
        assert len(set(flat)) == len(flat) == per * world and set(flat) <= set(range(n))
        if not shuffle:
            assert got[0] == list(range(0, per * world, world))
        else:
            perm = torch.randperm(n, generator=torch.Generator().manual_seed(3 + 2)).tolist()[:per * world]
            assert got == [perm[r::world] for r in range(world)]
            assert rank_indices(n, 0, world, True, seed=3, epoch=3) != got[0] or n <= 2   # set_epoch reshuffles
            assert rank_indices(n, 0, world, True, seed=3, epoch=2) == got[0]
