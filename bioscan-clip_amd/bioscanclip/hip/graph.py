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
No process-group collective sits inside a captured region: with world_size > 1 the step is a handful of per-tower graphs with
the collectives issued between them (``GraphedDistStep``).
"""
import os

import torch

# Capture mode (round 4, the GPUTEST_r03 abort).  hipEventQuery / hipStreamQuery are on HIP's list of calls that are illegal
# while ANY stream of the process is being captured in the default "global" mode -- from every thread.  ProcessGroupNCCL's
# watchdog thread polls the end events of the collectives it still lists (hipEventQuery, every ~100 ms): when such a poll
# landed inside a global-mode capture the query failed, the watchdog rethrew from its own thread and the process died with
# SIGABRT (ProcessGroupNCCL.cpp:2099, WorkNCCL::finishedGPUExecutionInternal).  torch.cuda.synchronize() completes the GPU work
# but does not make the watchdog drop its Work objects, so whether a poll hits the capture window was a race.  THE fix: every
# capture here is "thread_local" -- only the capturing thread is held to the restricted call set, the watchdog's queries stay
# legal (tests/test_90_dist_gpu.py::test_event_polling_thread_during_capture_is_legal makes the race deterministic).  Round 4 kept
# a 0.35 s sleep beside it ("long enough for the watchdog to retire its list"): it rested on a torch-internal poll interval and
# protected nothing that thread-local capture does not; it is gone (round 5), and "global" capture is refused while an NCCL process
# group exists instead of being left as a way back into the race.
CAPTURE_MODE = os.environ.get("BSCLIP_CAPTURE_MODE", "thread_local")


def quiesce_process_group():
    """Device idle when a capture begins (the warm-up steps' collectives have executed); refuses the capture mode in which the
    process group's watchdog thread would be held to the capture's restricted call set."""
    import torch.distributed as dist
    torch.cuda.synchronize()
    if CAPTURE_MODE == "global" and dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        raise RuntimeError("BSCLIP_CAPTURE_MODE=global with an NCCL process group: the group's watchdog thread polls events, which is "
                           "illegal during a global-mode capture from any thread (the round-3 abort); use the default thread_local")


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
            quiesce_process_group()
            self.captured_for = self._signature()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_MODE):
                self.loss = self._body()
            # the capture ran the host half of the step once (step counts) without executing anything; lr is delivered outside
            # the graph, before the replay that consumes it
            self.optimizer.stage_hyper()
            self.graph.replay()
            return self.loss
        self.optimizer.advance_host_state()
        self.graph.replay()
        return self.loss


class _Tower:
    __slots__ = ("name", "enc", "x", "stream", "gF", "gB", "emb", "z", "dz", "full", "flat", "work", "comm")


# BSCLIP_NATIVE_COMM=1: the W > 1 step's collectives through the C-ABI entry points (bsclip_allgather_embeddings /
# bsclip_allgather_labels / bsclip_allreduce_grads: csrc/comm.hip, hip/dist.py NativeComm) instead of torch.distributed -- one RCCL
# communicator and one communication stream PER TOWER (plus one for the labels), each collective ordered behind its tower's stream by
# an event and announcing its completion by an event.  What that removes: ProcessGroupNCCL's single stream (a gather issued behind the
# image tower's used to wait for the image tower: hence "shortest tower first" below), its watchdog thread, and the Work objects.
# Default off: torch.distributed (the same RCCL underneath) stays the product path until an 8-GPU node has run both
# (tests/test_90_dist_gpu.py holds the two paths bit-equal at world_size 1, the only size this box has).
NATIVE_COMM = os.environ.get("BSCLIP_NATIVE_COMM", "0") == "1"


class GraphedDistStep(GraphedStep):
    """The global-batch step (world_size > 1) as per-tower captured hipGraphs with the collectives issued eagerly BETWEEN them,
    each from the stream of the tower whose data it moves (round 4; round 3 replayed forward | loss + backward | AdamW serially
    and lost north_star's overlap):

        main stream     [all-gather labels] ............ wait(gathers) gL: loss over the gathered batch, dL/dz per modality .... wait(all-reduces) gC: fused AdamW
        tower stream k  gF[k]: zero this encoder's gradients, forward, l2norm -> [all-gather z_k] .... gB[k]: backward of tower k -> [all-reduce(SUM) flat grads k]

    A modality's all-gather is enqueued the moment its tower's forward graph is launched, ordered behind THAT tower only: it runs
    on the process group's stream beside the other towers' encoders (the image tower is launched first and finishes last; the
    DNA and text gathers hide under it).  Each encoder's flat-gradient all-reduce is enqueued behind its own backward graph and
    runs beside the other towers' backward.  The loss graph waits for the gathers, the optimizer graph for the all-reduces:
    stream waits, never host waits (RCCL backend).  Per step the host issues 2 T + 2 graph launches and 2 T + 1 collectives
    (T towers) instead of ~1 100 ctypes launches.  No RCCL call sits inside a captured region -- the collectives are ordinary
    ``torch.distributed`` calls, the path RCCL is exercised on everywhere.  Kernel for kernel and stream for stream the replays
    are the eager global-batch step: tests/test_90_dist_gpu.py holds the two to the same losses and parameters on two ranks,
    bit for bit."""

    def __init__(self, model, optimizer, criterion, warmup=2, group=None, native_comm=None):
        super().__init__(model, optimizer, criterion, warmup)
        self.group = group if group is not None else getattr(criterion, "group", None)
        self.towers = None
        self.gL = self.gC = None
        self.native = NATIVE_COMM if native_comm is None else bool(native_comm)
        self.label_comm = None
        # bench.py --gpus N: (start, end) event pairs on the main stream around the waits for the gathers / the all-reduces of each
        # replayed step AFTER the towers' own graphs have been waited for, i.e. the collective time the overlap did not hide
        self.profile_waits = False
        self.wait_events = []

    def collective_wait_ms(self):
        """(exposed all-gather wait, exposed all-reduce wait) in ms per profiled step (call after a device synchronisation)."""
        if not self.wait_events:
            return None
        g = sum(a.elapsed_time(b) for a, b, _, _ in self.wait_events) / len(self.wait_events)
        r = sum(c.elapsed_time(d) for _, _, c, d in self.wait_events) / len(self.wait_events)
        return g, r

    _ORDER = {"text": 0, "dna": 1, "image": 2}    # issue order of the collectives: by the towers' duration, shortest first

    def n_graphs(self):
        return 0 if self.towers is None else sum((t.gF is not None) + (t.gB is not None) for t in self.towers) + 2

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

    def _make_towers(self):
        from ..model.simple_clip import _tower_stream
        image, dna, text, _ = self.static
        dev = self.static[3].device
        out = []
        # the eager step's host order (SimpleCLIP.forward: image, DNA, text) and the eager step's streams
        for k, name, enc, x in ((1, "image", self.model.image_encoder, image), (0, "dna", self.model.dna_encoder, dna),
                                (2, "text", self.model.language_encoder, text)):
            if enc is None:
                continue
            t = _Tower()
            t.name, t.enc, t.x, t.stream = name, enc, x, _tower_stream(k, dev)
            t.gF = t.gB = t.emb = t.z = t.dz = t.full = t.flat = t.work = t.comm = None
            out.append(t)
        if len(out) < 2:
            raise ValueError("Too less element for calculating the contrastive loss.")
        return out

    def _capture(self):
        import torch.distributed as dist
        from . import dist as hdist
        from .functional import infonce, l2_normalize
        label = self.static[3]
        W, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        B = label.shape[0]
        self.row0 = rank * B
        dev = label.device
        self.labels_full = torch.empty(W * B, dtype=label.dtype, device=dev)
        towers = self._make_towers()
        mode = dict(capture_error_mode=CAPTURE_MODE)
        was_on = hdist._OVERLAP["on"]
        hdist._OVERLAP["on"] = False            # no collective is issued from inside a captured region
        try:
            quiesce_process_group()
            self.captured_for = self._signature()
            pool = None
            for t in towers:                    # forward of one tower, on that tower's stream
                eng = getattr(t.enc, "_engine", None)
                t.flat = getattr(eng, "flat", None)
                t.gF = torch.cuda.CUDAGraph()
                with torch.cuda.graph(t.gF, pool=pool, stream=t.stream, **mode):
                    if t.flat is not None:
                        t.flat.bind_grads()
                        t.flat.grad.zero_()     # optimizer.zero_grad() for this encoder
                    t.emb = l2_normalize(t.enc(t.x))
                pool = t.gF.pool() if pool is None else pool
                t.full = torch.empty(W * B, t.emb.shape[1], dtype=torch.float32, device=dev)
            self.gL = torch.cuda.CUDAGraph()    # loss over the gathered batch, own rows re-inserted; dL/dz of every modality
            with torch.cuda.graph(self.gL, pool=pool, **mode):
                for t in towers:
                    t.z = t.emb.detach().requires_grad_(t.emb.requires_grad)
                gathered = [hdist._InsertLocal.apply(t.z, t.full, self.row0) for t in towers]
                loss = infonce(gathered, self.labels_full, self.criterion.logit_scale, row0=self.row0, n_local=B)
                live = [t for t in towers if t.z.requires_grad]
                for t, g in zip(live, torch.autograd.grad(loss, [t.z for t in live])):
                    t.dz = g
                self.loss_buf.copy_(loss.detach())
            for t in towers:                    # backward of one tower, on that tower's stream
                if t.dz is None:
                    continue
                t.gB = torch.cuda.CUDAGraph()
                with torch.cuda.graph(t.gB, pool=pool, stream=t.stream, **mode):
                    t.emb.backward(t.dz)
            self.gC = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.gC, pool=pool, **mode):
                self.optimizer.step()
        finally:
            hdist._OVERLAP["on"] = was_on
        if self.native:                         # one communicator + stream per tower, one for the labels (ids travel through the group)
            from .dist import NativeComm
            for t in towers:
                t.comm = NativeComm(rank, W, group=self.group, device=dev)
            self.label_comm = NativeComm(rank, W, group=self.group, device=dev)
        self.towers = towers
        self.graph = self.gL                    # "captured" marker for the base class's bookkeeping

    def _replay(self):
        import torch.distributed as dist
        label = self.static[3]
        main = torch.cuda.current_stream()
        native = self.native
        prof = self.profile_waits
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if prof else None
        if native:
            dones = [self.label_comm.all_gather_into(self.labels_full, label.contiguous())]
        else:
            works = [dist.all_gather_into_tensor(self.labels_full, label.contiguous(), group=self.group, async_op=True)]
        for t in self.towers:                   # longest tower first
            t.stream.wait_stream(main)
            with torch.cuda.stream(t.stream):
                t.gF.replay()
        # The process group runs its collectives on ONE stream, in issue order: a gather issued behind the image tower's would wait
        # for the image tower (tools/dist_overlap_probe.py, round 4: the text tower's gather, ready at 3.8 ms, completed at 18.0).
        # So they are issued shortest tower first -- each with its tower's stream current, i.e. ordered behind that tower only.
        # (Native path: every tower has its own communicator and stream; the same order is kept so that both paths pair up alike.)
        for t in sorted(self.towers, key=lambda t: self._ORDER[t.name]):
            with torch.cuda.stream(t.stream):
                if native:
                    dones.append(t.comm.all_gather_into(t.full, t.emb.detach()))
                else:
                    works.append(dist.all_gather_into_tensor(t.full, t.emb.detach(), group=self.group, async_op=True))
        for t in self.towers:
            main.wait_stream(t.stream)
        if prof:
            ev[0].record(main)
        if native:
            for d in dones:
                main.wait_event(d)
        else:
            for w in works:
                w.wait()                        # RCCL: `main` waits for the collective's stream; gloo (tests): the host does
        if prof:
            ev[1].record(main)
        self.gL.replay()
        for t in self.towers:
            t.work = None
            if t.gB is None:
                continue
            t.stream.wait_stream(main)
            with torch.cuda.stream(t.stream):
                t.gB.replay()
        for t in sorted(self.towers, key=lambda t: self._ORDER[t.name]):    # shortest backward first, as above; same order on every rank
            if t.gB is not None and t.flat is not None:
                with torch.cuda.stream(t.stream):
                    if native:
                        t.work = t.comm.all_reduce_sum_(t.flat.grad)
                    else:
                        t.work = dist.all_reduce(t.flat.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        for t in self.towers:
            main.wait_stream(t.stream)
        if prof:
            ev[2].record(main)
        for t in self.towers:
            if t.work is not None:
                if native:
                    main.wait_event(t.work)
                else:
                    t.work.wait()
        if prof:
            ev[3].record(main)
            self.wait_events.append(tuple(ev))
        self.gC.replay()
        if hasattr(self.optimizer, "sync_updated_slices"):
            self.optimizer.sync_updated_slices()      # sharded optimizer state (full fine-tuning): owners broadcast their slices
        return self.loss_buf

    def _drop(self):
        if self.towers is not None:
            for c in [t.comm for t in self.towers] + [self.label_comm]:
                if c is not None:
                    c.close()
        self.label_comm = None
        self.graph = self.gL = self.gC = self.towers = None

    def __call__(self, image, dna, text, label):
        self._stage(image, dna, text, label)
        if self.warmup_left > 0:
            self.warmup_left -= 1
            self.loss = self._eager()
            return self.loss
        if self.graph is not None and self._signature() != self.captured_for:
            self._drop()
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
