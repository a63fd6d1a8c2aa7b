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
