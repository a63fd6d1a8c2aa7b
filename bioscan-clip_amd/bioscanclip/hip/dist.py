"""Data-parallel pieces of the global-batch step, one process per GPU over torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

  * ``gather_features_and_labels``: all-gather of each modality's [B,768] embeddings and of the labels.  Gathered
    remote rows are constants; the local slice is re-inserted so only it carries gradient -- the reference's
    ``gather_features(gather_with_grad=False, local_loss=False)`` variant (loss_func.py:84-89).  Payloads are tiny
    (0.79 MB f32 per modality at B=256), i.e. latency-bound: one collective per modality, issued on a side stream
    as soon as that modality's embedding exists, so it overlaps the next encoder's head GEMMs.
  * ``allreduce_grads``: one all-reduce(SUM) per flat trainable-gradient buffer (5.9-7.6 MB).  SUM, not mean: the
    loss is already the global mean and each rank back-propagates only its own rows' dLoss/dz.
  * ``broadcast_trainable``: the reference broadcasts every parameter tensor one by one (train_cl.py:29-31); here one
    flat buffer per encoder (frozen weights are identical on every rank by construction).
The functions are compute-agnostic (they move tensors, nothing else), so the world_size-2 gloo tests drive them on
CPU tensors.
"""
import torch
import torch.distributed as dist


class _InsertLocal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, local, gathered, row0):
        ctx.row0, ctx.n = row0, local.shape[0]
        out = gathered.clone() if gathered.requires_grad else gathered
        out[row0:row0 + local.shape[0]] = local
        return out

    @staticmethod
    def backward(ctx, g):
        return g[ctx.row0:ctx.row0 + ctx.n], None, None


_side_stream = {}
# Rehearsal knob: BSCLIP_FORCE_DIST=1 sends a world_size-1 job through the collectives as well, so that a one-GPU box
# exercises the real RCCL calls (all-gather on the side stream, flat all-reduce, broadcast) of the multi-GPU step.
_FORCE = __import__("os").environ.get("BSCLIP_FORCE_DIST", "0") == "1"


def _inactive(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size(group) == 1 and not _FORCE


def _comm_stream(device):
    if device.type != "cuda":
        return None
    s = _side_stream.get(device)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _side_stream[device] = s
    return s


def gather_features_and_labels(feats, label, group=None):
    """feats: list of [B, D] f32 (autograd); label: [B] int64.  Returns ([W*B, D] per modality, [W*B] labels, row0)."""
    if _inactive(group):
        return feats, label, 0
    W, rank = dist.get_world_size(group), dist.get_rank(group)
    B = feats[0].shape[0]
    row0 = rank * B
    dev = feats[0].device
    side = _comm_stream(dev)
    outs, works = [], []
    if side is not None:
        side.wait_stream(torch.cuda.current_stream(dev))
    ctxm = torch.cuda.stream(side) if side is not None else _null()
    with ctxm:
        for f in feats:
            full = torch.empty(W * B, f.shape[1], dtype=f.dtype, device=dev)
            works.append(dist.all_gather_into_tensor(full, f.detach().contiguous(), group=group, async_op=True))
            outs.append(full)
        labels = torch.empty(W * B, dtype=label.dtype, device=dev)
        works.append(dist.all_gather_into_tensor(labels, label.contiguous(), group=group, async_op=True))
    for w in works:
        w.wait()
    if side is not None:
        torch.cuda.current_stream(dev).wait_stream(side)
    gathered = [_InsertLocal.apply(f, full, row0) for f, full in zip(feats, outs)]
    return gathered, labels, row0


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def flat_buffers(model):
    """The FlatParams of every HIP encoder engine under ``model`` (built lazily by the first forward)."""
    out = []
    for m in model.modules():
        eng = getattr(m, "_engine", None)
        if eng is not None and hasattr(eng, "flat"):
            out.append(eng.flat)
    return out


def allreduce_grads(model_or_buffers, group=None):
    if _inactive(group):
        return
    bufs = model_or_buffers if isinstance(model_or_buffers, (list, tuple)) else [f.grad for f in flat_buffers(model_or_buffers)]
    works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group, async_op=True) for b in bufs]
    for w in works:
        w.wait()


def broadcast_trainable(model_or_buffers, src=0, group=None):
    if _inactive(group):
        return
    if isinstance(model_or_buffers, (list, tuple)):
        bufs = list(model_or_buffers)
    else:
        # one broadcast per flat trainable buffer once the engines exist; trainable tensors not (yet) re-homed into a flat
        # buffer -- i.e. before the first forward, when the reference broadcasts (train_cl.py:149) -- go one by one
        flats = flat_buffers(model_or_buffers)
        covered = {id(p) for f in flats for p in f.params}
        bufs = [f.data for f in flats] + [p.data for p in model_or_buffers.parameters()
                                          if p.requires_grad and id(p) not in covered]
    for b in bufs:
        dist.broadcast(b, src=src, group=group)
