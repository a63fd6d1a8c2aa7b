"""Data-parallel pieces of the global-batch step, one process per GPU over torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

  * ``gather_features_and_labels``: all-gather of each modality's [B,768] embeddings and of the labels.  Gathered
    remote rows are constants; the local slice is re-inserted so only it carries gradient -- the reference's
    ``gather_features(gather_with_grad=False, local_loss=False)`` variant (loss_func.py:84-89).  Payloads are tiny
    (0.79 MB f32 per modality at B=256), i.e. latency-bound: one collective per modality.
  * Overlap (``enable_overlap``, switched on by ``GlobalBatchContrastiveLoss``): ``SimpleCLIP.forward`` calls
    ``start_gather`` on each TOWER's stream right after that tower's ``l2_normalize`` is enqueued, so a modality's
    all-gather runs on RCCL's stream beside the other towers' encoders; the loss only waits for the handles.  Labels
    are gathered at the start of the step (``start_label_gather``).  In backward each encoder's flat gradient buffer
    is all-reduced from ``_EncoderFn.backward`` the moment that tower's kernels are enqueued (``start_allreduce``),
    beside the other tower's backward; ``allreduce_grads`` then only waits.  Collectives are issued from the host
    thread (forward) and from autograd's per-device thread (backward) in an order fixed by the model, identical on
    every rank.  One ``backward()`` per optimizer step is assumed (as in the reference loop, train_epoch.py:28-42).
  * ``allreduce_grads``: one all-reduce(SUM) per flat trainable-gradient buffer (5.9-7.6 MB).  SUM, not mean: the
    loss is already the global mean and each rank back-propagates only its own rows' dLoss/dz.
  * ``broadcast_parameters``: every parameter and buffer from rank 0, as the reference does (train_cl.py:29-31,149) --
    ranks that build a random-init or differently-seeded trunk would otherwise train different models;
    ``broadcast_trainable`` (one broadcast per flat buffer) is the cheap re-sync once the engines exist.
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


_OVERLAP = {"on": False, "group": None}
_PENDING_AR = []      # [(flat gradient buffer, work handle)] issued from _EncoderFn.backward, waited in allreduce_grads
_ISSUED_AR = []       # the gradient buffers in the order their all-reduces were issued this step (must be rank-independent)
_ISSUED_TAGS = []     # ... and what each of them is ("ViTEngine:1476096"): compared across the ranks by allreduce_grads
_AR_ORDER = {"sig": None}
_PENDING_LABELS = {}  # id(label tensor) -> (label tensor, gathered labels, work handle)


def enable_overlap(group=None, on=True):
    """Issue the per-modality all-gathers from the tower streams and the per-encoder gradient all-reduces from the
    encoder autograd nodes (see the module docstring).  Set by GlobalBatchContrastiveLoss; harmless without a process
    group (every hook checks ``overlap_active``)."""
    _OVERLAP["on"], _OVERLAP["group"] = bool(on), group
    _AR_ORDER["sig"] = None
    _PENDING_AR.clear()
    _PENDING_LABELS.clear()


def overlap_active():
    return _OVERLAP["on"] and not _inactive(_OVERLAP["group"])


def _all_gather_async(t, group):
    W = dist.get_world_size(group)
    full = torch.empty((W * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    return full, dist.all_gather_into_tensor(full, t.detach().contiguous(), group=group, async_op=True)


def start_gather(y):
    """Called on the stream that produced ``y`` ([B, D] embedding of one modality), right after its l2_normalize: the
    collective is ordered behind that stream only, the handle rides on the tensor until the loss picks it up."""
    if overlap_active():
        y._bsclip_gather = _all_gather_async(y, _OVERLAP["group"])
    return y


def start_label_gather(label):
    """Start of the step: labels are known before any encoder runs."""
    if overlap_active():
        _PENDING_LABELS.clear()
        _PENDING_LABELS[id(label)] = (label,) + _all_gather_async(label, _OVERLAP["group"])


def start_allreduce(flat, tag=""):
    """Called from an encoder's autograd node on that tower's stream once its whole backward is enqueued."""
    if overlap_active():
        if not _PENDING_AR:
            _ISSUED_AR.clear()
            _ISSUED_TAGS.clear()
        _ISSUED_AR.append(flat.grad)
        _ISSUED_TAGS.append(f"{tag}:{flat.grad.numel()}")
        _PENDING_AR.append((flat.grad, dist.all_reduce(flat.grad, op=dist.ReduceOp.SUM, group=_OVERLAP["group"],
                                                       async_op=True)))


def _finish(full, work):
    work.wait()  # CUDA: the current stream waits for RCCL's stream; CPU (gloo): blocks
    if full.is_cuda:
        full.record_stream(torch.cuda.current_stream(full.device))
    return full


def gather_features_and_labels(feats, label, group=None):
    """feats: list of [B, D] f32 (autograd); label: [B] int64.  Returns ([W*B, D] per modality, [W*B] labels, row0).
    Gathers already started by ``start_gather`` / ``start_label_gather`` are only waited for here."""
    if _inactive(group):
        return feats, label, 0
    rank = dist.get_rank(group)
    B = feats[0].shape[0]
    row0 = rank * B
    handles = []
    for f in feats:
        h = getattr(f, "_bsclip_gather", None)
        handles.append(h if h is not None else _all_gather_async(f, group))
    pend = _PENDING_LABELS.pop(id(label), None)
    lab = pend[1:] if pend is not None and pend[0] is label else _all_gather_async(label, group)
    outs = [_finish(*h) for h in handles]
    labels = _finish(*lab)
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


def _check_issue_order(group):
    """Collectives pair up across ranks by issue order alone, and the per-encoder all-reduces are issued from autograd's
    thread: the first overlapped step compares the order (which encoder, how many elements) across ALL ranks -- every rank makes
    that call, also one that issued nothing, and every rank sees every rank's list, so they agree on the verdict and fail together
    (a rank that raised alone left its peers blocked in the exchange).  Later steps compare with the first step locally:
    the order is a property of the model's autograd graph, which does not change; a rank that finds a change raises at once, and the
    launcher (torchrun / mp.spawn) tears the job down when one worker exits -- the peers do not wait for a collective verdict that
    would cost every step a host synchronisation."""
    sig = list(_ISSUED_TAGS)
    if _AR_ORDER["sig"] is None:
        theirs = [None] * dist.get_world_size(group)
        dist.all_gather_object(theirs, sig, group=group)
        if any(t != theirs[0] for t in theirs):
            raise RuntimeError(f"gradient all-reduces were issued in different orders on different ranks: {theirs}")
        _AR_ORDER["sig"] = sig
    elif sig != _AR_ORDER["sig"]:
        raise RuntimeError(f"gradient all-reduce issue order changed: {sig} vs {_AR_ORDER['sig']} on the first step")


def allreduce_grads(model_or_buffers, group=None):
    """SUM the flat trainable-gradient buffers over the ranks.  Buffers whose all-reduce was already started by
    ``start_allreduce`` (overlap mode) are only waited for."""
    if _inactive(group):
        return
    bufs = model_or_buffers if isinstance(model_or_buffers, (list, tuple)) else [f.grad for f in flat_buffers(model_or_buffers)]
    started = {b.data_ptr(): w for b, w in _PENDING_AR}
    if overlap_active() and (_AR_ORDER["sig"] is None or _PENDING_AR):   # the first overlapped step: on EVERY rank, pending or not
        if not _PENDING_AR:
            _ISSUED_TAGS.clear()
        _check_issue_order(group)
    _PENDING_AR.clear()
    works = [started.pop(b.data_ptr(), None) or dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group, async_op=True)
             for b in bufs]
    for w in list(started.values()) + works:
        w.wait()


def broadcast_parameters(model, src=0, group=None):
    """Reference ``broadcast_model`` (train_cl.py:29-31): every parameter (and buffer) of the model from rank ``src``.
    In-place copies bump the tensors' version counters, so engines that already packed the frozen weights repack."""
    if _inactive(group):
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def frozen_checksum(model):
    """float64 sum over every non-trainable floating-point tensor (on the tensors' device): a cheap cross-rank
    equality check of the frozen trunk."""
    tensors = [t for t in list(model.parameters()) + list(model.buffers())
               if not getattr(t, "requires_grad", False) and t.is_floating_point()]
    if not tensors:          # full fine-tuning (disable_lora): nothing is frozen, nothing to compare
        return None
    total = torch.zeros(1, dtype=torch.float64, device=tensors[0].device)
    for t in tensors:
        total += t.detach().double().sum()
    return total


def assert_frozen_in_sync(model, group=None):
    """Raise if the ranks hold different frozen weights (they would silently train different models)."""
    if _inactive(group):
        return
    mine = frozen_checksum(model)
    if mine is None:         # every rank takes this branch together: the set of frozen tensors is a property of the config
        return
    allv = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(allv, mine, group=group)
    vals = [v.item() for v in allv]
    if any(v != vals[0] for v in vals):
        raise RuntimeError(f"frozen weights differ across ranks: checksums {vals}")


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


class NativeComm:
    """The C-ABI collectives (``bsclip_comm_*`` / ``bsclip_allgather_*`` / ``bsclip_allreduce_grads``, include/bsclip.h) with
    their own RCCL communicator and communication stream -- what a caller without ``torch.distributed`` binds.  The 128-byte
    unique id travels from rank 0 through whatever channel the caller has (here: an existing ``torch.distributed`` group of any
    backend, or nothing at world_size 1).  The product path keeps ``torch.distributed`` (identical collectives, same RCCL
    underneath); this class is the drop-in for it and is exercised at world_size 1 on the one-GPU test box
    (tests/test_90_dist_gpu.py)."""

    def __init__(self, rank=0, world=1, group=None, device=None):
        import ctypes
        from . import lib as _l
        self._l, self._ct = _l, ctypes
        self.rank, self.world = int(rank), int(world)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        h = _l.load()
        n = h.bsclip_comm_unique_id_bytes()
        buf = ctypes.create_string_buffer(n)
        if self.rank == 0:
            _l.check(h.bsclip_comm_unique_id(buf))
        if self.world > 1:
            t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            if dist.get_backend(group) == "nccl":
                t = t.to(self.device)
            dist.broadcast(t, src=0, group=group)
            buf = ctypes.create_string_buffer(bytes(t.cpu().tolist()), n)
        comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _l.check(h.bsclip_comm_init(ctypes.byref(comm), buf, self.rank, self.world))
        self.comm = comm
        self.stream = torch.cuda.Stream(device=self.device)

    def _events(self):
        """(event recorded now on the current stream = the payload's producer, event the collective records when done)."""
        ready, done = torch.cuda.Event(), torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        return ready, done

    def _ev(self, e):
        return self._ct.c_void_p(e.cuda_event)

    def all_gather(self, local):
        """local: contiguous f32 [B, D] or int64 [B] on this device.  Returns (gathered, done_event); the collective is
        ordered behind the CURRENT stream's work and runs on the communicator's stream."""
        h = self._l.load()
        local = local.contiguous()
        out = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        ready, done = self._events()
        fn = h.bsclip_allgather_embeddings if local.dtype == torch.float32 else h.bsclip_allgather_labels
        if local.dtype not in (torch.float32, torch.int64):
            raise TypeError("NativeComm.all_gather: f32 embeddings or int64 labels")
        self._l.check(fn(self.comm, self._ct.c_void_p(local.data_ptr()), self._ct.c_void_p(out.data_ptr()), local.numel(),
                         self._ct.c_void_p(self.stream.cuda_stream), self._ev(ready), self._ev(done)))
        local.record_stream(self.stream)
        out.record_stream(self.stream)
        return out, done

    def all_gather_into(self, out, local):
        """The same into a caller-owned buffer ``out`` [world * B, ...] (a captured graph reads it: no allocation per step).  Returns
        the done event; nothing is synchronised on the host."""
        h = self._l.load()
        if local.dtype not in (torch.float32, torch.int64) or out.dtype != local.dtype:
            raise TypeError("NativeComm.all_gather_into: f32 embeddings or int64 labels, same dtype in and out")
        if not (local.is_contiguous() and out.is_contiguous()) or out.numel() != self.world * local.numel():
            raise ValueError("NativeComm.all_gather_into: contiguous buffers, out = world x local")
        ready, done = self._events()
        fn = h.bsclip_allgather_embeddings if local.dtype == torch.float32 else h.bsclip_allgather_labels
        self._l.check(fn(self.comm, self._ct.c_void_p(local.data_ptr()), self._ct.c_void_p(out.data_ptr()), local.numel(),
                         self._ct.c_void_p(self.stream.cuda_stream), self._ev(ready), self._ev(done)))
        local.record_stream(self.stream)
        out.record_stream(self.stream)
        return done

    def all_reduce_sum_(self, flat):
        """In-place SUM over ranks of a flat f32 buffer; returns the done event."""
        if flat.dtype != torch.float32 or not flat.is_contiguous():
            raise TypeError("NativeComm.all_reduce_sum_: contiguous f32 buffer")
        ready, done = self._events()
        self._l.check(self._l.load().bsclip_allreduce_grads(self.comm, self._ct.c_void_p(flat.data_ptr()), flat.numel(),
                                                            self._ct.c_void_p(self.stream.cuda_stream), self._ev(ready),
                                                            self._ev(done)))
        flat.record_stream(self.stream)
        return done

    def close(self):
        if self.comm is not None:
            self.stream.synchronize()
            self._l.check(self._l.load().bsclip_comm_destroy(self.comm))
            self.comm = None
