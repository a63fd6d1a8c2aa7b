"""Fused AdamW over the engines' flat trainable buffers (reference: ``optim.AdamW(model.parameters(), lr)`` at
scripts/train_cl.py:158 -- torch defaults betas (0.9, 0.999), eps 1e-8, weight_decay 0.01, applied to LoRA and head
weights *and biases*; frozen parameters have no gradient and are skipped)."""
import torch

from . import ops


class FusedAdamW(torch.optim.Optimizer):
    """Drop-in for ``torch.optim.AdamW``: same constructor, ``param_groups[0]['lr']`` honoured every step (so
    torch LR schedulers work).  Parameters that live in an engine ``FlatParams`` buffer are updated with one
    ``bsclip_adamw_step`` launch per buffer; any other trainable tensor gets its own launch of the same kernel."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._flat_state = {}
        self._flats = []
        self._dev_hyper = False
        self._shard = None        # (rank, world, group): optimizer-state sharding over the ranks (shard_state)

    def shard_state(self, group=None):
        """Multi-rank full fine-tuning (every rank would otherwise hold the full 2 x 4 B per parameter of AdamW
        moments: 3.2 GB for I+D+T): each rank keeps the moments of ONE contiguous slice of every flat buffer and updates only
        that slice; after the update every slice is broadcast from its owner (``sync_updated_slices``, W small collectives per
        flat buffer -- they work on every backend and between graph replays).  Gradients are still all-reduced over the whole
        buffer (hip/dist.py), so the update itself is the unsharded one, element for element."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            self._shard = None
            return self
        if self._flat_state:
            raise RuntimeError("FusedAdamW.shard_state: call before the first step (moments are already allocated unsharded)")
        self._shard = (dist.get_rank(group), dist.get_world_size(group), group)
        return self

    def _slice_of(self, f, rank=None):
        """[lo, hi) of flat buffer ``f`` owned by ``rank`` (default: this rank): equal 4-element-aligned slices, the last one short."""
        n = f.data.numel()
        if self._shard is None:
            return 0, n
        r, W, _ = self._shard
        r = r if rank is None else rank
        per = (-(-n // W) + 3) // 4 * 4
        return min(r * per, n), min((r + 1) * per, n)

    def sync_updated_slices(self):
        """Broadcast every rank's freshly updated parameter slice to the others.  Eager: called by ``step()`` outside a capture
        and by ``GraphedDistStep`` after its optimizer graph."""
        if self._shard is None:
            return
        import torch.distributed as dist
        _, W, group = self._shard
        works = []
        for f in self._flats:
            if not f.valid():
                continue
            for r in range(W):
                lo, hi = self._slice_of(f, r)
                if hi > lo:
                    works.append(dist.broadcast(f.data[lo:hi], src=dist.get_global_rank(group, r) if group is not None else r,
                                                group=group, async_op=True))
        for w in works:
            w.wait()

    def enable_device_hyper(self, on=True):
        """Read (lr, step) from device memory inside the kernel (``bsclip_adamw_step_dev``) instead of passing them as launch
        arguments: required when the step is captured into a hipGraph (arguments freeze at capture).  Each flat buffer owns a
        device pair ``[lr (f32), step (uint32)]``.  No in-flight node ever reads host memory: the step word is advanced on the
        device (eagerly it is set by a fill kernel carrying the value as a launch argument; under capture a
        ``bsclip_counter_add`` node advances it on every replay), and lr reaches the device through ``stage_hyper()`` -- a
        stream-ordered fill issued at call time, OUTSIDE the graph, before the replay that consumes it."""
        self._dev_hyper = bool(on)

    def _group_of(self, f):
        for g in self.param_groups:
            if any(p is f.params[0] for p in g["params"]):
                return g
        return self.param_groups[0]

    def stage_hyper(self):
        """Enqueue the current lr of every flat buffer's own param group into its device word (a fill kernel whose value is a
        launch argument: stream-ordered behind the previous step, nothing for the host to overwrite while it is in flight).
        Called by ``step()`` when it is not being captured and by ``GraphedStep`` before each replay."""
        for f in self._flats:
            st = self._flat_state.get(id(f))
            if st is not None and "hyper" in st:
                lr = float(self._group_of(f)["lr"])
                if st.get("lr_staged") != lr:
                    st["hyper"].view(torch.float32)[0:1].fill_(lr)
                    st["lr_staged"] = lr

    def advance_host_state(self):
        """Host half of one optimizer step, for a graph replay: the host's step counts follow the device words (which the
        replayed graph advances itself) and the current lr is staged."""
        for f in self._flats:
            st = self._flat_state.get(id(f))
            if st is not None and "hyper" in st:
                st["step"] += 1
        self.stage_hyper()

    def _state_for(self, f):
        """Adam moments of one flat buffer.  Keyed by the PARAMETERS it holds (they outlive an engine rebuild: a checkpoint
        loaded mid-run, a precision switch), not by the buffer object; every parameter's slice is also published in
        ``self.state[p]`` as views (``exp_avg``, ``exp_avg_sq``, ``step``) so ``state_dict()`` / ``load_state_dict()`` carry
        the moments like torch.optim.AdamW's do."""
        key = tuple(id(p) for p in f.params)
        st = self._flat_state.get(id(f))
        if self._shard is not None:
            lo, hi = self._slice_of(f)
            if st is None or st["m"].numel() != hi - lo or st.get("key") != key:
                prev = next((v for v in self._flat_state.values() if v.get("key") == key and v.get("slice") == (lo, hi)), None)
                st = {"m": torch.zeros(hi - lo, dtype=f.data.dtype, device=f.data.device),
                      "v": torch.zeros(hi - lo, dtype=f.data.dtype, device=f.data.device), "step": 0, "key": key, "slice": (lo, hi)}
                if prev is not None:                  # engine rebuilt (checkpoint loaded mid-run): the optimisation continues
                    st["m"].copy_(prev["m"])
                    st["v"].copy_(prev["v"])
                    st["step"] = prev["step"]
                self._flat_state = {k: v for k, v in self._flat_state.items() if v.get("key") != key}
                self._flat_state[id(f)] = st
            return st
        if st is None or st["m"].numel() != f.data.numel() or st.get("key") != key:
            prev = next((v for v in self._flat_state.values() if v.get("key") == key and v["m"].numel() == f.data.numel()), None)
            st = {"m": torch.zeros_like(f.data), "v": torch.zeros_like(f.data), "step": 0, "key": key}
            if prev is not None:                      # engine rebuilt: the optimisation continues, it does not restart
                st["m"].copy_(prev["m"])
                st["v"].copy_(prev["v"])
                st["step"] = prev["step"]
            else:                                     # moments restored by load_state_dict() before the buffer existed
                for p, o in zip(f.params, f.offsets):
                    ps = self.state.get(p)
                    if ps and "exp_avg" in ps and ps["exp_avg"].numel() == p.numel():
                        st["m"][o:o + p.numel()].copy_(ps["exp_avg"].reshape(-1).to(st["m"].device))
                        st["v"][o:o + p.numel()].copy_(ps["exp_avg_sq"].reshape(-1).to(st["v"].device))
                        st["step"] = max(st["step"], int(ps.get("step", 0)))
            self._flat_state = {k: v for k, v in self._flat_state.items() if v.get("key") != key}
            self._flat_state[id(f)] = st
            for p, o in zip(f.params, f.offsets):
                self.state[p] = {"step": st["step"], "exp_avg": st["m"][o:o + p.numel()].view(p.shape),
                                 "exp_avg_sq": st["v"][o:o + p.numel()].view(p.shape)}
        return st

    def state_dict(self):
        if self._shard is not None:
            raise NotImplementedError("FusedAdamW: the moments are sharded over the ranks (shard_state); the reference checkpoints "
                                      "the model only (train_cl.py:219-220)")
        for f in self._flats:                         # the per-parameter step counts follow the flat buffer's
            st = self._flat_state.get(id(f))
            if st is not None:
                for p in f.params:
                    if p in self.state:
                        self.state[p]["step"] = st["step"]
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat_state = {}                         # next step(): moments are copied from self.state into the flat buffers

    def attach(self, model):
        """Tell the optimizer which engines' flat buffers exist (called by train_epoch after the first forward)."""
        from .dist import flat_buffers
        self._flats = flat_buffers(model)

    def needs_attach(self):
        """True until the engines' flat buffers are known, and again after an engine was rebuilt (checkpoint loaded mid-run)."""
        return not self._flats or not all(f.valid() for f in self._flats)

    def zero_grad(self, set_to_none: bool = True):
        handled = set()
        for f in self._flats:
            if f.valid():
                f.bind_grads()
                f.grad.zero_()
                handled.update(id(p) for p in f.params)
        for group in self.param_groups:
            for p in group["params"]:
                if id(p) in handled or p.grad is None:
                    continue
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        handled = set()
        in_opt = {id(p): g for g in self.param_groups for p in g["params"]}
        for f in self._flats:
            if not f.valid() or not all(id(p) in in_opt for p in f.params):
                continue
            g = in_opt[id(f.params[0])]
            st = self._state_for(f)
            f.bind_grads()
            st["step"] += 1
            lo, hi = self._slice_of(f)
            if hi <= lo:
                handled.update(id(p) for p in f.params)
                continue
            fdata, fgrad = (f.data, f.grad) if self._shard is None else (f.data[lo:hi], f.grad[lo:hi])
            if self._dev_hyper:
                capturing = torch.cuda.is_current_stream_capturing()
                if "hyper" not in st:
                    if capturing:
                        raise RuntimeError("FusedAdamW: run one eager step with device-side (lr, step) before capturing")
                    st["hyper"] = torch.zeros(2, dtype=torch.int32, device=f.data.device)   # [lr as f32 bits, step as uint32]
                if capturing:
                    ops.counter_add(st["hyper"][1:2], 1)          # a node of the graph: every replay advances the step word
                else:
                    st["hyper"][1:2].fill_(st["step"])             # value travels as a launch argument
                    lr = float(g["lr"])
                    if st.get("lr_staged") != lr:
                        st["hyper"].view(torch.float32)[0:1].fill_(lr)
                        st["lr_staged"] = lr
                ops.adamw_step_dev(fdata, fgrad, st["m"], st["v"], st["hyper"], g["betas"][0], g["betas"][1], g["eps"],
                                   g["weight_decay"])
            else:
                ops.adamw_step(fdata, fgrad, st["m"], st["v"], g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                               g["weight_decay"], st["step"])
            handled.update(id(p) for p in f.params)
        for group in self.param_groups:
            for p in group["params"]:
                if id(p) in handled or p.grad is None:
                    continue
                if self._dev_hyper:
                    raise RuntimeError("FusedAdamW: device-side (lr, step) needs every trainable tensor in an engine's flat buffer")
                if not (p.is_cuda and p.dtype == torch.float32 and p.data.is_contiguous() and p.grad.is_contiguous()):
                    raise RuntimeError("FusedAdamW: parameters must be contiguous f32 GPU tensors (no CPU path)")
                st = self.state[p]
                if not st:
                    st["m"], st["v"], st["step"] = torch.zeros_like(p.data), torch.zeros_like(p.data), 0
                st["step"] += 1
                ops.adamw_step(p.data.view(-1), p.grad.view(-1), st["m"].view(-1), st["v"].view(-1), group["lr"],
                               group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"], st["step"])
        if self._shard is not None and not torch.cuda.is_current_stream_capturing():
            self.sync_updated_slices()
        return loss
