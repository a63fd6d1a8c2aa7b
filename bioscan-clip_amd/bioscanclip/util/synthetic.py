"""Synthetic stand-in for the HDF5 data layer (reference ``bioscanclip/util/dataset.py``, out of scope: SURVEY 2.1 #4).

Yields batches in the reference's 7-tuple layout (dataset.py:267-275): ``(processid, image [B,3,224,224] f32 in [0,1),
dna_tokens [B,133] i64, input_ids, token_type_ids, attention_mask [B,20] i64 (or None), label [B] i64)``.
"""
import torch


class SyntheticCLIPLoader:
    def __init__(self, batch_size, steps, with_text=False, seed=1234, rank=0, world_size=1):
        self.batch_size, self.steps, self.with_text = batch_size, steps, with_text
        self.seed, self.rank, self.world = seed, rank, world_size

    def __len__(self):
        return self.steps

    def __iter__(self):
        B = self.batch_size
        for s in range(self.steps):
            g = torch.Generator().manual_seed(self.seed + 1000 * s + self.rank)
            image = torch.rand(B, 3, 224, 224, generator=g)
            dna = torch.randint(3, 1027, (B, 133), generator=g)
            dna[:, 0] = 0  # the literal <MASK>=0 prefix of get_sequence_pipeline (dna_encoder.py:33)
            ids = tt = am = None
            if self.with_text:
                ids = torch.randint(1000, 30522, (B, 20), generator=g)
                lens = torch.randint(6, 21, (B,), generator=g)
                am = (torch.arange(20)[None] < lens[:, None]).long()
                ids = ids * am
                ids[:, 0] = 101
                ids[torch.arange(B), lens - 1] = 102
                tt = torch.zeros_like(ids)
            label = torch.arange(B) + (s * self.world + self.rank) * B
            pid = [f"SYN{self.rank}_{s}_{i}" for i in range(B)]
            yield pid, image, dna, ids, tt, am, label


class SyntheticEvalLoader(SyntheticCLIPLoader):
    """Evaluation-split stand-in: same tensors, but ``label`` is the dict of taxonomy-name lists that the reference's
    evaluation loaders collate (``for_training=False``, dataset.py:267-275) and ``convert_label_dict_to_list_of_dict`` expects."""

    def __init__(self, batch_size, steps, with_text=False, seed=4321):
        super().__init__(batch_size, steps, with_text=with_text, seed=seed)

    def __iter__(self):
        for s, (pid, image, dna, ids, tt, am, label) in enumerate(super().__iter__()):
            if ids is None:  # the reference's 7-tuple always carries token tensors
                ids = tt = am = torch.zeros(len(pid), 20, dtype=torch.int64)
            sp = [int(v) % 11 for v in label]
            lab = {"order": [f"o{v % 2}" for v in sp], "family": [f"f{v % 3}" for v in sp],
                   "genus": [f"g{v % 5}" for v in sp], "species": [f"s{v}" for v in sp]}
            yield pid, image, dna, ids, tt, am, lab


class SyntheticRawLoader(SyntheticCLIPLoader):
    """``dataset=synthetic_raw``: the stored form of the data -- decoded uint8 images of assorted sizes and nucleotide strings of
    assorted lengths -- pushed through the GPU input pipeline (bioscanclip/util/gpu_pipeline.py: the reference's Dataset_for_CL
    transforms, dataset.py:171-181, and get_sequence_pipeline, dna_encoder.py:25-35) inside the loader, so a training run
    exercises it end to end.  Yields the same 7-tuple as SyntheticCLIPLoader, with image / dna already on the device."""

    SIZES = [(256, 256), (288, 256), (256, 341), (320, 320)]

    def __init__(self, batch_size, steps, with_text=False, seed=1234, rank=0, world_size=1, for_training=True, device="cuda"):
        super().__init__(batch_size, steps, with_text=with_text, seed=seed, rank=rank, world_size=world_size)
        from bioscanclip.util.gpu_pipeline import GpuAugment
        self.augment = GpuAugment(for_training=for_training, seed=seed + 17 * rank)
        self.device = device

    def __iter__(self):
        from bioscanclip.util.gpu_pipeline import tokenize_barcodes
        B = self.batch_size
        for s, (pid, _, _, ids, tt, am, label) in enumerate(super().__iter__()):
            g = torch.Generator().manual_seed(self.seed + 7919 * s + self.rank)
            images = []
            for i in range(B):
                h, w = self.SIZES[int(torch.randint(0, len(self.SIZES), (1,), generator=g))]
                images.append(torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8))
            seqs = []
            for i in range(B):
                L = int(torch.randint(500, 720, (1,), generator=g))
                seqs.append("".join("ACGT"[int(c)] for c in torch.randint(0, 4, (L,), generator=g)))
            image, _ = self.augment(images, device=self.device)
            dna = tokenize_barcodes(seqs, device=self.device)
            yield pid, image, dna, ids, tt, am, label
