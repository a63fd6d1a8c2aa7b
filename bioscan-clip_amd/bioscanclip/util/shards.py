"""Pre-decoded shards: the data side of the input pipeline (reference bioscanclip/util/dataset.py:97-275).

The reference keeps a split as datasets of one HDF5 file (DATA.md:23-35) and decodes a JPEG per ``__getitem__`` on a DataLoader
worker.  h5py and a JPEG decoder are absent from this image, and at >= 6 k images/s per GPU a CPU decode per sample is what
starves the step, so the stored form here is what the GPU pipeline consumes directly -- one directory per split:

    meta.json                       {"format": "bsclip-shard", "version": 1, "n", "split", "dataset", "text_len"}
    images.bin                      decoded uint8 H x W x 3 pixels of every sample, back to back
    image_index.npy                 int64 [n, 3]: byte offset into images.bin, H, W
    barcodes.bin                    the nucleotide strings (ASCII), back to back      (HDF5 "barcode")
    barcode_offsets.npy             int64 [n + 1]
    language_tokens_input_ids.npy   int64 [n, text_len]                               (HDF5 datasets of the same names)
    language_tokens_token_type_ids.npy, language_tokens_attention_mask.npy
    processid.npy                   bytes [n]   (HDF5 "processid" for BIOSCAN-5M, "image_file" for BIOSCAN-1M: dataset.py:243-246)
    labels.npy                      int64 [n]   training labels (dataset.py:135-142: the sample index unless a label array is given)
    order.npy, family.npy, genus.npy, species.npy   bytes [n]  (get_array_of_label_dicts, dataset.py:51-62: evaluation labels)

Every array is memory-mapped; nothing is loaded whole.  ``convert_hdf5_split`` writes this layout from the reference's HDF5 file
wherever h5py and PIL exist (not in this image: documented, exercised by nothing here).

``ShardLoader`` yields the reference's 7-tuple ``(processid, image, dna, input_ids, token_type_ids, attention_mask, label)``
(dataset.py:267-275) with ``image`` f32 [B, 3, 224, 224] and ``dna`` int64 [B, 133] already on the device:
  * sample order: ``prepare()`` (dataset.py:41-48) = ``DistributedSampler(num_replicas, rank, shuffle, drop_last=True)`` -- the
    same torch class, so ``set_epoch`` reshuffles exactly as the reference's loader does -- batched without dropping the tail;
  * a producer thread gathers batch k+1 from the memory-mapped arrays into PINNED staging buffers while batch k trains, then, on
    a side HIP stream, copies it to the device and runs the two per-sample transforms there (``GpuAugment``: Resize 256 ->
    RandomResizedCrop 224 -> flips -> rotation, or Resize -> CenterCrop for evaluation; ``tokenize_barcodes``); the consumer's
    stream waits on the batch's event, the host never blocks on the copy.  Three staging slots; a slot's pinned buffers are refilled
    only after its previous copies have executed; the tensors handed to the consumer are fresh allocations of the side stream
    (``record_stream`` keeps the allocator from recycling them under the consumer).
"""
import json
import os
import queue
import threading

import numpy as np
import torch

FORMAT, VERSION = "bsclip-shard", 1
_TAXA = ("order", "family", "genus", "species")
_TEXT = ("language_tokens_input_ids", "language_tokens_token_type_ids", "language_tokens_attention_mask")


def write_shard(path, images, barcodes, input_ids, token_type_ids, attention_mask, processid, labels=None, taxonomy=None,
                split="train", dataset="bioscan_1m"):
    """images: sequence of uint8 [H, W, 3] arrays (sizes may differ); barcodes: sequence of str; the three text arrays
    int [n, text_len]; processid: sequence of str; labels: int [n] (default: the sample index, dataset.py:139);
    taxonomy: dict order/family/genus/species -> sequence of str (default: empty strings)."""
    os.makedirs(path, exist_ok=True)
    n = len(barcodes)
    assert len(processid) == n and len(input_ids) == n
    index = np.zeros((n, 3), dtype=np.int64)
    off, seen = 0, 0
    with open(os.path.join(path, "images.bin"), "wb") as f:
        for i, im in enumerate(images):                       # any iterable: images are streamed to disk one at a time
            im = np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
            assert im.ndim == 3 and im.shape[2] == 3, "images must be decoded uint8 H x W x 3"
            index[i] = (off, im.shape[0], im.shape[1])
            f.write(im.tobytes())
            off += im.size
            seen += 1
    assert seen == n, f"{seen} images for {n} barcodes"
    np.save(os.path.join(path, "image_index.npy"), index)
    offs = np.zeros(n + 1, dtype=np.int64)
    with open(os.path.join(path, "barcodes.bin"), "wb") as f:
        for i, s in enumerate(barcodes):
            b = s.encode("ascii", "replace") if isinstance(s, str) else bytes(s)
            f.write(b)
            offs[i + 1] = offs[i] + len(b)
    np.save(os.path.join(path, "barcode_offsets.npy"), offs)
    for name, arr in zip(_TEXT, (input_ids, token_type_ids, attention_mask)):
        np.save(os.path.join(path, name + ".npy"), np.asarray(arr, dtype=np.int64).reshape(n, -1))
    np.save(os.path.join(path, "processid.npy"), np.asarray([str(p).encode() for p in processid], dtype=np.bytes_))
    np.save(os.path.join(path, "labels.npy"), np.arange(n, dtype=np.int64) if labels is None else np.asarray(labels, dtype=np.int64))
    for t in _TAXA:
        vals = [""] * n if taxonomy is None else list(taxonomy[t])
        np.save(os.path.join(path, t + ".npy"), np.asarray([str(v).encode() for v in vals], dtype=np.bytes_))
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump({"format": FORMAT, "version": VERSION, "n": n, "split": split, "dataset": dataset,
                   "text_len": int(np.asarray(input_ids).reshape(n, -1).shape[1])}, f)


def is_shard(path):
    return isinstance(path, str) and os.path.isfile(os.path.join(path, "meta.json"))


class Shard:
    """Memory-mapped view of one split directory."""

    def __init__(self, path):
        with open(os.path.join(path, "meta.json")) as f:
            self.meta = json.load(f)
        if self.meta.get("format") != FORMAT or self.meta.get("version") != VERSION:
            raise ValueError(f"{path}: not a {FORMAT} v{VERSION} directory (meta.json says {self.meta})")
        self.path, self.n = path, int(self.meta["n"])
        ld = lambda name: np.load(os.path.join(path, name + ".npy"), mmap_mode="r")
        self.image_index = ld("image_index")
        self.images = np.memmap(os.path.join(path, "images.bin"), dtype=np.uint8, mode="r")
        self.barcode_offsets = ld("barcode_offsets")
        self.barcodes = np.memmap(os.path.join(path, "barcodes.bin"), dtype=np.uint8, mode="r")
        self.input_ids, self.token_type_ids, self.attention_mask = (ld(t) for t in _TEXT)
        self.processid, self.labels = ld("processid"), ld("labels")
        self.taxonomy = {t: ld(t) for t in _TAXA}
        if not (len(self.image_index) == len(self.labels) == len(self.processid) == self.n and len(self.barcode_offsets) == self.n + 1):
            raise ValueError(f"{path}: array lengths disagree with meta.json n={self.n}")

    def __len__(self):
        return self.n

    def image(self, i):
        off, h, w = (int(v) for v in self.image_index[i])
        return self.images[off:off + h * w * 3].reshape(h, w, 3)

    def barcode(self, i):
        return bytes(self.barcodes[int(self.barcode_offsets[i]):int(self.barcode_offsets[i + 1])]).decode("ascii")

    def label_dicts(self, idx):
        """The evaluation labels of ``get_array_of_label_dicts`` for the given samples."""
        return [{t: self.taxonomy[t][i].decode() for t in _TAXA} for i in idx]


class _Range(torch.utils.data.Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


def rank_indices(n, rank, world_size, shuffle, seed, epoch):
    """The sample order of the reference's ``prepare()`` for one rank and epoch: torch's own DistributedSampler with
    ``drop_last=True`` (dataset.py:42)."""
    sampler = torch.utils.data.DistributedSampler(_Range(n), num_replicas=world_size, rank=rank, shuffle=shuffle, seed=seed,
                                                  drop_last=True)
    sampler.set_epoch(epoch)
    return list(iter(sampler))


class ShardLoader:
    """Iterable over one split; see the module docstring.  ``for_training`` selects the augmentation chain and index labels
    (dataset.py:135-144), otherwise Resize -> CenterCrop and the taxonomy dictionaries as labels."""

    SLOTS = 3

    def __init__(self, shard, batch_size, rank=0, world_size=1, shuffle=False, seed=0, for_training=True, with_text=True,
                 device="cuda", augment_seed=None):
        from bioscanclip.util.gpu_pipeline import GpuAugment
        self.shard = Shard(shard) if isinstance(shard, str) else shard
        self.batch_size, self.rank, self.world_size = int(batch_size), rank, world_size
        self.shuffle, self.seed, self.epoch = shuffle, seed, 0
        self.for_training, self.with_text, self.device = for_training, with_text, torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.augment = GpuAugment(for_training=for_training, seed=(seed + 17 * rank) if augment_seed is None else augment_seed)
        self.stream = torch.cuda.Stream(device=self.device)
        self._slots = [dict(cap_img=0, cap_dna=0) for _ in range(self.SLOTS)]
        self.last_params = None      # the augmentation draws of the most recently yielded batch (tests replay them on the CPU oracle)
        self.last_indices = None

    def set_epoch(self, epoch):
        """``DistributedSampler.set_epoch``: a different shuffle per epoch (and the reference's behaviour when it is never called:
        the same order every epoch)."""
        self.epoch = int(epoch)

    def indices(self):
        return rank_indices(len(self.shard), self.rank, self.world_size, self.shuffle, self.seed, self.epoch)

    def __len__(self):
        n = len(self.indices())
        return (n + self.batch_size - 1) // self.batch_size

    # ------------------------------------------------------------------------------------------------ producer side
    def _stage(self, slot, idx):
        """Host half of one batch: gather the samples into the slot's pinned buffers and draw the augmentation parameters."""
        from bioscanclip.util.gpu_pipeline import _resized_size
        sh = self.shard
        rows = sh.image_index[np.asarray(idx)]
        nbytes = int((rows[:, 1] * rows[:, 2]).sum()) * 3
        dna_len = int(sum(int(sh.barcode_offsets[i + 1] - sh.barcode_offsets[i]) for i in idx))
        if slot["cap_img"] < nbytes:
            slot["cap_img"] = int(nbytes * 1.25) + 4096
            slot["img_host"] = torch.empty(slot["cap_img"], dtype=torch.uint8).pin_memory()
            with torch.cuda.stream(self.stream):    # the staging buffer belongs to the stream that writes and reads it: a block
                # of the default stream's pool could still be in use by queued kernels of the training step
                slot["img_dev"] = torch.empty(slot["cap_img"], dtype=torch.uint8, device=self.device)
        if slot["cap_dna"] < dna_len + 1:
            slot["cap_dna"] = int(dna_len * 1.25) + 1024
            slot["dna_host"] = torch.empty(slot["cap_dna"], dtype=torch.uint8).pin_memory()
            with torch.cuda.stream(self.stream):
                slot["dna_dev"] = torch.empty(slot["cap_dna"], dtype=torch.uint8, device=self.device)
        img_np, dna_np = slot["img_host"].numpy(), slot["dna_host"].numpy()
        sizes, off, doffs = [], 0, [0]
        for (o, h, w), i in zip(rows, idx):
            nb = int(h) * int(w) * 3
            img_np[off:off + nb] = sh.images[int(o):int(o) + nb]
            off += nb
            sizes.append((int(h), int(w)))
            b0, b1 = int(sh.barcode_offsets[i]), int(sh.barcode_offsets[i + 1])
            dna_np[doffs[-1]:doffs[-1] + (b1 - b0)] = sh.barcodes[b0:b1]
            doffs.append(doffs[-1] + (b1 - b0))
        sel = np.asarray(idx)
        B = len(idx)
        if "meta_host" not in slot:   # small per-batch arrays share one pinned block per slot, allocated once (pinning is a
            T = int(self.shard.meta["text_len"])   # driver call that can synchronise the device: never per batch)
            slot["meta_host"] = torch.empty((self.batch_size + 1) + self.batch_size + 3 * self.batch_size * T,
                                            dtype=torch.int64).pin_memory()
        mh, T = slot["meta_host"], int(sh.meta["text_len"])
        off_h = mh[:B + 1]
        off_h.copy_(torch.tensor(doffs, dtype=torch.int64))
        host = {"sizes": sizes, "nbytes": nbytes, "dna_offsets": off_h, "n_dna": doffs[-1],
                "params": [self.augment.sample(*_resized_size(h, w, self.augment.resize_to)) for h, w in sizes],
                "processid": [p.decode() for p in sh.processid[sel]]}
        if self.for_training:
            lab = mh[self.batch_size + 1:self.batch_size + 1 + B]
            lab.copy_(torch.from_numpy(np.ascontiguousarray(sh.labels[sel])))
            host["label"] = lab
        else:
            host["label"] = sh.label_dicts(idx)
        if self.with_text:
            base = 2 * self.batch_size + 1
            host["text"] = []
            for k, a in enumerate((sh.input_ids, sh.token_type_ids, sh.attention_mask)):
                t = mh[base + k * self.batch_size * T:base + k * self.batch_size * T + B * T].view(B, T)
                t.copy_(torch.from_numpy(np.ascontiguousarray(a[sel])))
                host["text"].append(t)
        return host

    def _enqueue(self, slot, host):
        """Device half, on the loader's side stream: H2D copies + the two transforms; records the slot's ready event."""
        from bioscanclip.hip import ops
        B = len(host["sizes"])
        with torch.cuda.stream(self.stream):
            slot["img_dev"][:host["nbytes"]].copy_(slot["img_host"][:host["nbytes"]], non_blocking=True)
            n_dna = host["n_dna"]
            slot["dna_dev"][:max(n_dna, 1)].copy_(slot["dna_host"][:max(n_dna, 1)], non_blocking=True)
            off_dev = host["dna_offsets"].to(self.device, non_blocking=True)
            image, _ = self.augment.run_packed(slot["img_dev"], host["sizes"], host["params"], device=self.device)
            dna = torch.empty(B, 660 // 5 + 1, dtype=torch.int64, device=self.device)
            ops.kmer_tokenize(slot["dna_dev"], off_dev, B, 660, 5, dna)
            out = {"image": image, "dna": dna, "processid": host["processid"], "params": host["params"]}
            if self.with_text:
                out["text"] = [t.to(self.device, non_blocking=True) for t in host["text"]]
            else:
                out["text"] = [None, None, None]
            out["label"] = host["label"].to(self.device, non_blocking=True) if self.for_training else host["label"]
            ev = torch.cuda.Event()
            ev.record(self.stream)
            out["ready"] = ev
        slot["out"] = out
        return out

    def __iter__(self):
        idx = self.indices()
        batches = [idx[i:i + self.batch_size] for i in range(0, len(idx), self.batch_size)]
        q = queue.Queue(maxsize=self.SLOTS - 1)
        dev = self.device

        def producer():
            try:
                torch.cuda.set_device(dev)
                for k, b in enumerate(batches):
                    slot = self._slots[k % self.SLOTS]
                    if "out" in slot:
                        # the slot's pinned buffers are about to be refilled: its previous H2D copies (and the kernels that read
                        # its device staging buffers) must have run -- three batches ago, so this never waits in practice
                        slot["out"]["ready"].synchronize()
                    host = self._stage(slot, b)
                    q.put((k, b, self._enqueue(slot, host)))
                q.put(None)
            except BaseException as exc:   # noqa: BLE001 - surface producer failures in the consumer
                q.put(exc)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        while True:
            item = q.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            k, b, out = item
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(out["ready"])
            for t in (out["image"], out["dna"], out["label"], *[x for x in out["text"] if x is not None]):
                if torch.is_tensor(t):       # every tensor allocated on the loader's stream and consumed on the caller's
                    t.record_stream(cur)
            self.last_params, self.last_indices = out["params"], b
            yield (out["processid"], out["image"], out["dna"], out["text"][0], out["text"][1], out["text"][2], out["label"])
        th.join()


def convert_hdf5_split(hdf5_path, split, out_dir, dataset="bioscan_1m", labels=None):
    """Write one split of the reference's HDF5 file (DATA.md:23-35; datasets image / image_mask / barcode /
    language_tokens_* / processid | image_file / order / family / genus / species, read as ``Dataset_for_CL`` reads them,
    dataset.py:219-265) as a shard directory.  Needs h5py and PIL, which this image does not have: run it where the data lives."""
    try:
        import io
        import h5py
        from PIL import Image
    except ImportError as exc:   # pragma: no cover - neither package exists in this image
        raise ImportError("convert_hdf5_split needs h5py and Pillow (absent from the build image); run it on the data host") from exc
    g = h5py.File(hdf5_path, "r", libver="latest")[split]                     # pragma: no cover
    n = len(g["barcode"])                                                     # pragma: no cover
    pid_key = "processid" if dataset == "bioscan_5m" else "image_file"        # pragma: no cover

    def images():                                                             # pragma: no cover
        for i in range(n):
            enc = g["image"][i].astype(np.uint8)[:g["image_mask"][i]]
            yield np.asarray(Image.open(io.BytesIO(enc.tobytes())).convert("RGB"), dtype=np.uint8)
    write_shard(out_dir, images(), [b.decode("utf-8") for b in g["barcode"][:]],                         # pragma: no cover
                g["language_tokens_input_ids"][:], g["language_tokens_token_type_ids"][:], g["language_tokens_attention_mask"][:],
                [p.decode("utf-8") for p in g[pid_key][:]], labels=labels,
                taxonomy={t: [v.decode("utf-8") for v in g[t][:]] for t in _TAXA}, split=split, dataset=dataset)
