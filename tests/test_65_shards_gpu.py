"""SURVEY 8f-3, data side, on the GPU: the shard loader yields the reference's 7-tuple (dataset.py:267-275) with the two per-sample
transforms done on the device behind a pinned, double-buffered H2D path -- checked sample by sample against the CPU oracle pipeline
(oracle/pipeline.py) on the committed tiny shard, through train_cl.py, and for the overlap plumbing (events, slot reuse)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = os.path.join(ROOT, "tests", "golden", "tiny_shard")

from helpers import rel_err  # noqa: E402
from oracle import pipeline as opipe  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.mark.parametrize("for_training,world,rank,with_text", [(True, 1, 0, True), (True, 2, 1, False), (False, 1, 0, True)])
def test_loader_matches_the_cpu_pipeline(for_training, world, rank, with_text):
    from bioscanclip.util import shards
    sh = shards.Shard(TINY)
    ld = shards.ShardLoader(TINY, batch_size=5, rank=rank, world_size=world, shuffle=for_training, seed=11, for_training=for_training,
                            with_text=with_text)
    for epoch in (0, 1):
        ld.set_epoch(epoch)
        want_idx = shards.rank_indices(24, rank, world, for_training, 11, epoch)
        seen = []
        nb = 0
        for pid, image, dna, ids, tt, am, label in ld:
            idx = ld.last_indices
            B = len(idx)
            nb += 1
            seen += idx
            assert image.shape == (B, 3, 224, 224) and image.dtype == torch.float32 and image.is_cuda
            assert dna.shape == (B, 133) and dna.dtype == torch.int64 and dna.is_cuda
            torch.cuda.synchronize()
            for k, i in enumerate(idx):
                ref = opipe.augment(torch.from_numpy(np.array(sh.image(i))), ld.last_params[k])
                assert rel_err(image[k].cpu(), ref) < 2e-5, (epoch, k, i)
            assert torch.equal(dna.cpu(), opipe.kmer_tokenize([sh.barcode(i) for i in idx]))
            assert pid == [f"TINY{i:04d}" for i in idx]
            if with_text:
                sel = np.asarray(idx)
                assert torch.equal(ids.cpu(), torch.from_numpy(np.array(sh.input_ids[sel])))
                assert torch.equal(tt.cpu(), torch.from_numpy(np.array(sh.token_type_ids[sel])))
                assert torch.equal(am.cpu(), torch.from_numpy(np.array(sh.attention_mask[sel])))
            else:
                assert ids is None and tt is None and am is None
            if for_training:
                assert torch.equal(label.cpu(), torch.tensor(idx))            # dataset.py:139: the sample index
                assert any(p["angle"] != 0.0 for p in ld.last_params)
            else:
                assert label == sh.label_dicts(idx)                           # dataset.py:144: taxonomy dictionaries
                assert all(p["angle"] == 0.0 and not p["hflip"] and p["box"][2:] == (224, 224) for p in ld.last_params)
        assert seen == want_idx and nb == len(ld) == -(-len(want_idx) // 5)


def test_slots_are_not_overwritten_while_the_consumer_reads_them():
    """Three staging slots, the producer running ahead on its own stream: a batch handed to the consumer must still hold its
    values after the consumer has queued slow work behind it and the producer has moved on."""
    from bioscanclip.util import shards
    ld = shards.ShardLoader(TINY, batch_size=2, shuffle=False, for_training=False, with_text=True)
    big = torch.randn(4096, 4096, device="cuda")
    kept = []
    for pid, image, dna, ids, tt, am, label in ld:
        for _ in range(3):
            big = big @ big * 1e-4                 # slow consumer work queued on the consumer's stream
        kept.append((image.clone(), dna.clone(), ids.clone()))    # reads the batch AFTER that work, in stream order
    torch.cuda.synchronize()
    ref = shards.ShardLoader(TINY, batch_size=2, shuffle=False, for_training=False, with_text=True)
    for (s, d, i), (pid, image, dna, ids, tt, am, label) in zip(kept, ref):
        torch.cuda.synchronize()
        assert torch.equal(s, image) and torch.equal(d, dna) and torch.equal(i, ids)


@pytest.mark.timeout(900)
def test_train_cl_reads_a_shard_directory(tmp_path, capsys):
    """``train_cl.py dataset=<shard dir>``: two epochs over the tiny shard (24 samples, batch 8 -> 3 steps per epoch, a new shuffle
    per epoch), I+D+T at full depth, the default (captured-graph) launch path."""
    scripts = os.path.join(ROOT, "bioscan-clip_amd", "scripts")
    sys.path.insert(0, scripts)
    import train_cl
    torch.manual_seed(5)
    losses = train_cl.main(["model_config=lora_vit_lora_barcode_bert_lora_bert_ssl", "model_config.batch_size=8", "model_config.epochs=2",
                            f"dataset={TINY}", "debug_flag=true", f"project_root_path={tmp_path}"])
    capsys.readouterr()
    assert len(losses) == 2 and all(l == l and 0 < l < 20 for l in losses)
    with pytest.raises(NotImplementedError, match="convert_hdf5_split"):
        train_cl.main(["model_config=lora_vit_lora_barcode_bert_ssl", "dataset=/data/BIOSCAN_1M/split_data/BioScan_data_in_splits.hdf5"])
