"""Row 8e on the real HIP path: two ranks sharing one GPU (gloo moves the tensors; the one-GPU test box cannot host an
RCCL group) must reproduce the single-process step on the concatenated batch -- loss and trainable gradients -- through
the product's own `GlobalBatchContrastiveLoss` (all-gather + local-rows gradient) and flat-gradient all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from helpers import rel_err  # noqa: E402
from oracle import synth  # noqa: E402

NODROP = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)


def _build():
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    from bioscanclip.model.image_encoder import LoRA_ViT_timm
    from bioscanclip.model.simple_clip import SimpleCLIP
    img = LoRA_ViT_timm(arch.VisionTransformerParams(depth=2), r=4, num_classes=768)
    dna = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=2, **NODROP)), r=4,
                            num_classes=768)
    model = SimpleCLIP(img, dna, None)
    model.load_state_dict(synth.synth_state_dict(synth.shapes_of(model), seed=51))
    return model.cuda().train()


def _flat_grads(model):
    return torch.cat([p.grad.reshape(-1) for _, p in sorted(model.named_parameters()) if p.requires_grad]).cpu()


def _worker(rank, world, port, B, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bioscanclip.hip import dist as hdist
    from bioscanclip.model.loss_func import GlobalBatchContrastiveLoss
    model = _build()
    image, dna, _, label = synth.synth_batch(world * B, seed=9, dup_labels=True)
    sl = slice(rank * B, (rank + 1) * B)
    crit = GlobalBatchContrastiveLoss(torch.nn.CrossEntropyLoss(), 1 / 0.07)
    io, do, _ = model(image[sl].cuda(), dna[sl].cuda(), None)
    loss = crit(io, do, None, label[sl].cuda())
    loss.backward()
    hdist.allreduce_grads(model)
    torch.cuda.synchronize()
    torch.save({"loss": loss.detach().cpu(), "flat": _flat_grads(model)}, os.path.join(tmp, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_match_single_process(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world, B = 2, 4
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    from bioscanclip.model.loss_func import ContrastiveLoss
    model = _build()
    image, dna, _, label = synth.synth_batch(world * B, seed=9, dup_labels=True)
    io, do, _ = model(image.cuda(), dna.cuda(), None)
    loss = ContrastiveLoss(torch.nn.CrossEntropyLoss(), 1 / 0.07)(io, do, None, label.cuda())
    loss.backward()
    flat = _flat_grads(model)
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"))
    assert abs(r0["loss"].item() - loss.item()) < 1e-5 * abs(loss.item())
    assert abs(r1["loss"].item() - loss.item()) < 1e-5 * abs(loss.item())
    assert torch.equal(r0["flat"], r1["flat"])
    assert rel_err(r0["flat"], flat) < 2e-3
