"""The training step as ONE hipGraph ("HIP streams and graphs instead of a tracing compiler").

Eagerly a step is ~560 ctypes launches: 8.6 ms of host time (measured at B=8, where the step IS the enqueue time; 2.7 ms for
one graph replay).  ``GraphedStep`` runs the step's own Python once under stream capture -- zero_grad, the towers on their streams, loss, backward through autograd,
fused AdamW -- and replays the resulting graph; per step the host then does three things: stage the learning rate (a fill
kernel enqueued ahead of the replay; the step count lives on the device and is advanced by a node of the graph, so no
in-flight node reads host memory the host could rewrite), launch the graph, and (if the caller wants it) read the loss.  What makes that legal:
  * no launch argument changes from step to step: dropout masks come from a per-engine device step word advanced by a node of
    the graph (``bsclip_set_dropout_step`` / ``bsclip_counter_add``), AdamW's lr and step from device memory
    (``bsclip_adamw_step_dev``), inputs from static buffers;
  * nothing in the step synchronises or allocates outside the capture's private pool (workspaces are built by the eager
    warm-up steps; the library never allocates).
One process group collective inside a capture is avoided on purpose: with world_size > 1 the step stays eager (the towers'
all-gathers and the per-encoder all-reduces are issued from Python).
"""
import torch


class GraphedStep:
    def __init__(self, model, optimizer, criterion, warmup=2):
        self.model, self.optimizer, self.criterion = model, optimizer, criterion
        self.warmup_left = max(1, int(warmup))
        self.graph = None
        self.static = None
        self.loss = None
        self.loss_buf = None
        self.captured_for = None
        if not hasattr(optimizer, "enable_device_hyper"):
            raise TypeError("GraphedStep needs FusedAdamW (device-side lr / step)")
        optimizer.enable_device_hyper(True)

    def _body(self):
        image, dna, text, label = self.static
        self.optimizer.zero_grad()
        loss = self.criterion(*self.model(image, dna, text), label)
        loss.backward()
        if self.optimizer.needs_attach():
            self.optimizer.attach(self.model)
        self.optimizer.step()
        # the value leaves the step through a buffer allocated before the capture (tensors created under capture live in
        # the graph's private pool; nothing outside a replay should depend on them)
        if self.loss_buf is None:
            self.loss_buf = torch.zeros((), dtype=torch.float32, device=loss.device)
        self.loss_buf.copy_(loss.detach())
        return self.loss_buf

    def _stage(self, image, dna, text, label):
        new = (image, dna, text, label)
        if self.static is None:
            clone = lambda t: None if t is None else t.clone()
            self.static = (clone(image), clone(dna), None if text is None else {k: v.clone() for k, v in text.items()},
                           clone(label))
            return
        for dst, src in zip(self.static, new):
            if dst is None:
                continue
            if isinstance(dst, dict):
                for k in dst:
                    if dst[k].data_ptr() != src[k].data_ptr():
                        dst[k].copy_(src[k], non_blocking=True)
            elif dst.data_ptr() != src.data_ptr():
                if dst.shape != src.shape:
                    raise ValueError(f"GraphedStep was captured for batch shape {tuple(dst.shape)}, got {tuple(src.shape)}")
                dst.copy_(src, non_blocking=True)

    def _signature(self):
        """What the captured graph holds raw pointers into: each engine, its workspace (replaced when a forward of another
        batch size runs in between -- the evaluation phase -- and then freed), its flat parameter buffer (replaced when a
        checkpoint is loaded mid-run).  A replay after any of them changed would write into freed memory."""
        sig = []
        for m in self.model.modules():
            eng = getattr(m, "_engine", None)
            if eng is not None:
                flat = getattr(eng, "flat", None)
                ws = getattr(eng, "ws", None)
                sig.append((id(eng), None if ws is None else ws["gen"], 0 if flat is None else flat.data.data_ptr()))
        return tuple(sig)

    def accepts(self, image, dna, text, label):
        """False for a batch the captured graph cannot take (another shape / another set of modalities than the static
        buffers hold): the caller enqueues that step eagerly."""
        if self.static is None:
            return True
        for dst, src in zip(self.static, (image, dna, text, label)):
            if (dst is None) != (src is None):
                return False
            if dst is None:
                continue
            if isinstance(dst, dict):
                if set(dst) != set(src) or any(dst[k].shape != src[k].shape for k in dst):
                    return False
            elif dst.shape != src.shape or dst.dtype != src.dtype:
                return False
        return True

    def __call__(self, image, dna, text, label):
        """One optimisation step; returns the loss as a device scalar (no host synchronisation)."""
        self._stage(image, dna, text, label)
        if self.warmup_left > 0:          # eager steps: engines, workspaces, flat buffers, optimizer state come to exist
            self.warmup_left -= 1
            self.loss = self._body()
            return self.loss
        if self.graph is not None and self._signature() != self.captured_for:
            self.graph = None                 # an engine / workspace / flat buffer was replaced: one eager step, then re-capture
            self.loss = self._body()
            return self.loss
        if self.graph is None:
            torch.cuda.synchronize()
            self.captured_for = self._signature()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.loss = self._body()
            # the capture ran the host half of the step once (step counts) without executing anything; lr is delivered outside
            # the graph, before the replay that consumes it
            self.optimizer.stage_hyper()
            self.graph.replay()
            return self.loss
        self.optimizer.advance_host_state()
        self.graph.replay()
        return self.loss


class GraphedDistStep(GraphedStep):
    """The global-batch step (world_size > 1) as THREE captured hipGraphs with the collectives issued eagerly between them:

        [all-gather labels]  graph A: zero_grad + the towers' forward  ->  [all-gather each modality's embeddings]
        graph B: loss over the gathered batch (own rows re-inserted) + backward through every encoder  ->  [all-reduce (SUM)
        each encoder's flat gradient buffer]  ->  graph C: fused AdamW.

    Eagerly a rank spends ~35 ms of host time per step on ~1 100 ctypes launches (the GPU step is 41 ms at local batch 256:
    the rank is host-bound next to 7 others); here it issues 3 graph launches and 5-7 collectives.  No RCCL call sits inside a
    captured region -- the collectives are ordinary ``torch.distributed`` calls on the process group between replays, so the
    path is the one RCCL is exercised on everywhere -- at the price of the overlap the eager path has (a modality's 0.8 MB
    all-gather beside the other towers' encoders, a 6 MB all-reduce beside the other tower's backward: tens of microseconds).
    Kernel for kernel the replays are the eager step: tests/test_dist_gpu.py holds the two to the same losses and parameters
    on two ranks."""

    def __init__(self, model, optimizer, criterion, warmup=2, group=None):
        super().__init__(model, optimizer, criterion, warmup)
        self.group = group if group is not None else getattr(criterion, "group", None)
        self.gA = self.gB = self.gC = None

    def _eager(self):
        image, dna, text, label = self.static
        from . import dist as hdist
        self.optimizer.zero_grad()
        if hasattr(self.criterion, "prefetch_labels"):
            self.criterion.prefetch_labels(label)
        loss = self.criterion(*self.model(image, dna, text), label)
        loss.backward()
        hdist.allreduce_grads(self.model, self.group)
        if self.optimizer.needs_attach():
            self.optimizer.attach(self.model)
        self.optimizer.step()
        if self.loss_buf is None:
            self.loss_buf = torch.zeros((), dtype=torch.float32, device=loss.device)
        self.loss_buf.copy_(loss.detach())
        return self.loss_buf

    def _capture(self):
        import torch.distributed as dist
        from . import dist as hdist
        from .functional import infonce
        image, dna, text, label = self.static
        W, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        B = label.shape[0]
        self.row0 = rank * B
        dev = label.device
        self.labels_full = torch.empty(W * B, dtype=label.dtype, device=dev)
        was_on = hdist._OVERLAP["on"]
        hdist._OVERLAP["on"] = False            # no collective is issued from inside a captured region
        try:
            torch.cuda.synchronize()
            self.captured_for = self._signature()
            self.gA = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.gA):
                self.optimizer.zero_grad()
                outs = self.model(image, dna, text)
            self.emb = [o for o in outs if o is not None]
            self.full = [torch.empty(W * B, e.shape[1], dtype=torch.float32, device=dev) for e in self.emb]
            self.gB = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.gB, pool=self.gA.pool()):
                gathered = [hdist._InsertLocal.apply(e, f, self.row0) for e, f in zip(self.emb, self.full)]
                loss = infonce(gathered, self.labels_full, self.criterion.logit_scale, row0=self.row0, n_local=B)
                loss.backward()
                self.loss_buf.copy_(loss.detach())
            self.flats = [f for f in hdist.flat_buffers(self.model)]
            self.gC = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.gC, pool=self.gA.pool()):
                self.optimizer.step()
        finally:
            hdist._OVERLAP["on"] = was_on
        self.graph = self.gA                    # "captured" marker for the base class's bookkeeping

    def _replay(self):
        import torch.distributed as dist
        label = self.static[3]
        lw = dist.all_gather_into_tensor(self.labels_full, label.contiguous(), group=self.group, async_op=True)
        self.gA.replay()
        works = [dist.all_gather_into_tensor(f, e.detach(), group=self.group, async_op=True) for e, f in zip(self.emb, self.full)]
        for w in [lw] + works:
            w.wait()
        self.gB.replay()
        works = [dist.all_reduce(f.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for f in self.flats]
        for w in works:
            w.wait()
        self.gC.replay()
        if hasattr(self.optimizer, "sync_updated_slices"):
            self.optimizer.sync_updated_slices()      # sharded optimizer state (full fine-tuning): owners broadcast their slices
        return self.loss_buf

    def __call__(self, image, dna, text, label):
        self._stage(image, dna, text, label)
        if self.warmup_left > 0:
            self.warmup_left -= 1
            self.loss = self._eager()
            return self.loss
        if self.graph is not None and self._signature() != self.captured_for:
            self.graph = self.gA = self.gB = self.gC = None
            self.loss = self._eager()
            return self.loss
        if self.graph is None:
            self._capture()                     # ran the host half of one step (step counts) without executing anything ...
            self.optimizer.stage_hyper()
            self.loss = self._replay()          # ... the first replay executes it
            return self.loss
        self.optimizer.advance_host_state()
        self.loss = self._replay()
        return self.loss
