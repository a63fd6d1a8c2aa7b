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
        group0 = self.param_groups[0]
        in_opt = {id(p): g for g in self.param_groups for p in g["params"]}
        for f in self._flats:
            if not f.valid() or not all(id(p) in in_opt for p in f.params):
                continue
            g = in_opt[id(f.params[0])]
            st = self._flat_state.get(id(f))
            if st is None or st["m"].numel() != f.data.numel():
                st = {"m": torch.zeros_like(f.data), "v": torch.zeros_like(f.data), "step": 0}
                self._flat_state[id(f)] = st
            f.bind_grads()
            st["step"] += 1
            ops.adamw_step(f.data, f.grad, st["m"], st["v"], g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                           g["weight_decay"], st["step"])
            handled.update(id(p) for p in f.params)
        for group in self.param_groups:
            for p in group["params"]:
                if id(p) in handled or p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.data.is_contiguous() and p.grad.is_contiguous()):
                    raise RuntimeError("FusedAdamW: parameters must be contiguous f32 GPU tensors (no CPU path)")
                st = self.state[p]
                if not st:
                    st["m"], st["v"], st["step"] = torch.zeros_like(p.data), torch.zeros_like(p.data), 0
                st["step"] += 1
                ops.adamw_step(p.data.view(-1), p.grad.view(-1), st["m"].view(-1), st["v"].view(-1), group["lr"],
                               group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"], st["step"])
        return loss
