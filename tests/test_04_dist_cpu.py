"""world_size-2 gloo tests (CPU) of the data-parallel step semantics:

    W-rank step with all-gathered embeddings + local-rows gradient + all-reduce(SUM) of the trainable gradients
      ==  single-process step on the concatenated batch   (loss equal, gradients equal).

The collectives under test are the product's (`bioscanclip.hip.dist`); the arithmetic is the CPU oracle, because the
HIP kernels cannot run here -- the distributed logic is compute-agnostic by construction."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import refcpu, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiny_state():
    from bioscanclip.model import arch
    from bioscanclip.model.dna_encoder import LoRA_barcode_bert
    from bioscanclip.model.language_encoder import LoRA_bert
    dna = LoRA_barcode_bert(arch.BertForMaskedLMParams(arch.barcode_bert_config(num_hidden_layers=1)), r=4, num_classes=768)
    txt = LoRA_bert(arch.BertModelParams(arch.bert_small_config(num_hidden_layers=1)), r=4, num_classes=768)
    shapes = {"dna_encoder." + k: v for k, v in synth.shapes_of(dna).items()}
    shapes.update({"language_encoder." + k: v for k, v in synth.shapes_of(txt).items()})
    return synth.synth_state_dict(shapes, seed=41)


def _forward(sd, dna, text):
    return (refcpu.l2_normalize(refcpu.barcode_bert_encoder(sd, dna)),
            refcpu.l2_normalize(refcpu.bert_text_encoder(sd, text)))


def _worker(rank, world, port, B, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from bioscanclip.hip import dist as hdist
    state = refcpu.StepState(_tiny_state())
    _, dna, text, label = synth.synth_batch(world * B, seed=9, with_text=True, dup_labels=True)
    sl = slice(rank * B, (rank + 1) * B)
    zd, zt = _forward(state.sd, dna[sl], {k: v[sl] for k, v in text.items()})
    # overlap mode, as the product runs it: the label gather starts at the top of the step, each modality's gather as soon
    # as its embedding exists (SimpleCLIP.forward -> start_gather), the loss only waits for the handles
    hdist.enable_overlap()
    local_label = label[sl].contiguous()
    hdist.start_label_gather(local_label)
    zd, zt = hdist.start_gather(zd), hdist.start_gather(zt)
    assert hasattr(zd, "_bsclip_gather") and hdist._PENDING_LABELS
    gathered, labels, row0 = hdist.gather_features_and_labels([zd, zt], local_label)
    assert not hdist._PENDING_LABELS
    assert row0 == rank * B and labels.tolist() == label.tolist()
    loss = refcpu.contrastive_loss(None, gathered[0], gathered[1], labels)
    loss.backward()
    flat = torch.cat([state.sd[k].grad.reshape(-1) for k in state.train_keys])
    # the encoder node starts its flat buffer's all-reduce itself (start_allreduce); allreduce_grads then only waits for it
    # and reduces whatever was not started (second buffer)
    class _Flat:
        grad = flat
    other = torch.full((5,), float(rank + 1))
    hdist.start_allreduce(_Flat)
    assert len(hdist._PENDING_AR) == 1
    hdist.allreduce_grads([flat, other])
    assert not hdist._PENDING_AR and (other == 3).all()
    hdist.enable_overlap(on=False)
    # the same collectives without overlap mode give the same gathered batch
    g2, l2, _ = hdist.gather_features_and_labels([zd.detach().clone(), zt.detach().clone()], local_label.clone())
    assert torch.equal(g2[0], gathered[0].detach()) and torch.equal(l2, labels)
    # broadcast of the flat trainable buffer (train_cl.py:29-31 semantics)
    buf = torch.full((7,), float(rank))
    hdist.broadcast_trainable([buf], src=0)
    assert (buf == 0).all()
    # before the first forward there are no flat buffers: a model's trainable tensors are broadcast one by one, the frozen
    # ones are left alone (they are identical by construction)
    from bioscanclip.model import arch
    from bioscanclip.model.language_encoder import LoRA_bert
    torch.manual_seed(100 + rank)
    txt = LoRA_bert(arch.BertModelParams(arch.bert_small_config(num_hidden_layers=1)), r=4, num_classes=768)
    frozen_before = {k: v.clone() for k, v in txt.named_parameters() if not v.requires_grad}
    hdist.broadcast_trainable(txt, src=0)
    torch.save({k: v.detach().clone() for k, v in txt.named_parameters() if v.requires_grad},
               os.path.join(tmp, f"trainable{rank}.pt"))
    assert all(torch.equal(v, frozen_before[k]) for k, v in txt.named_parameters() if not v.requires_grad)
    # train_cl.broadcast_model (reference train_cl.py:29-31,149): the ranks built DIFFERENT random trunks (seed 100 + rank);
    # the checksum guard sees it, broadcasting every parameter and buffer from rank 0 repairs it
    with pytest.raises(RuntimeError, match="frozen weights differ"):
        hdist.assert_frozen_in_sync(txt)
    hdist.broadcast_parameters(txt, src=0)
    hdist.assert_frozen_in_sync(txt)
    torch.save({k: v.detach().clone() for k, v in txt.state_dict().items()}, os.path.join(tmp, f"all{rank}.pt"))
    torch.save({"loss": loss.detach(), "flat": flat}, os.path.join(tmp, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_global_batch_step_equals_single_process(tmp_path):
    world, B = 2, 6
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    # single-process reference on the concatenated batch
    state = refcpu.StepState(_tiny_state())
    _, dna, text, label = synth.synth_batch(world * B, seed=9, with_text=True, dup_labels=True)
    zd, zt = _forward(state.sd, dna, text)
    loss = refcpu.contrastive_loss(None, zd, zt, label)
    loss.backward()
    flat = torch.cat([state.sd[k].grad.reshape(-1) for k in state.train_keys])
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"))
    assert torch.allclose(r0["loss"], loss.detach(), rtol=1e-6, atol=0)
    assert torch.allclose(r1["loss"], loss.detach(), rtol=1e-6, atol=0)
    assert torch.equal(r0["flat"], r1["flat"])
    t0, t1 = (torch.load(os.path.join(str(tmp_path), f"trainable{r}.pt")) for r in (0, 1))
    assert len(t0) >= 6 and all(torch.equal(t0[k], t1[k]) for k in t0)
    a0, a1 = (torch.load(os.path.join(str(tmp_path), f"all{r}.pt")) for r in (0, 1))
    assert len(a0) > 20 and all(torch.equal(a0[k], a1[k]) for k in a0)   # frozen trunk identical after broadcast_model
    err = ((r0["flat"] - flat).norm() / flat.norm()).item()
    assert err < 1e-5, err


def test_single_process_passthrough():
    """Without a process group the helpers are identities (N=1 bench path)."""
    from bioscanclip.hip import dist as hdist
    z = [torch.randn(4, 8), torch.randn(4, 8)]
    lab = torch.arange(4)
    g, l, row0 = hdist.gather_features_and_labels(z, lab)
    assert g[0] is z[0] and l is lab and row0 == 0
    hdist.allreduce_grads([torch.zeros(3)])
    hdist.broadcast_trainable([torch.zeros(3)])
