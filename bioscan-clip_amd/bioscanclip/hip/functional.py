"""Autograd nodes for the two small ops outside the encoders: F.normalize and the contrastive loss."""
import torch

from . import ops

F32 = torch.float32


class _L2Normalize(torch.autograd.Function):
    """``F.normalize(x, p=2, dim=-1)`` (reference simple_clip.py:34,47,49) on the HIP kernel."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        inv = torch.empty(x.shape[0], dtype=F32, device=x.device)
        ops.l2norm_fwd(x, y, inv)
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        dx = torch.empty_like(y)
        ops.l2norm_bwd(y, inv, dy.contiguous(), dx)
        return dx


def l2_normalize(x):
    if not x.is_cuda:
        raise RuntimeError("bioscanclip: tensors must live on the GPU (no CPU compute path)")
    return _L2Normalize.apply(x.to(F32))


class _InfoNCE(torch.autograd.Function):
    """ContrastiveLoss.forward (reference loss_func.py:29-54): loss and dLoss/dz come out of one fused call; the
    backward of this node only scales the stored gradients by the incoming scalar."""

    @staticmethod
    def forward(ctx, labels, scale, row0, n_local, workspace, *zs):
        zs = [z.contiguous() for z in zs]
        N = zs[0].shape[0]
        n_local = N if n_local is None else n_local
        loss = torch.zeros(1, dtype=F32, device=zs[0].device)
        need_grad = any(ctx.needs_input_grad[5:])
        dzs = [torch.empty(n_local, z.shape[1], dtype=F32, device=z.device) for z in zs] if need_grad else None
        ops.infonce_fwd_bwd(zs, labels, scale, loss, dzs, row0=row0, n_local=n_local, workspace=workspace)
        ctx.dzs, ctx.row0, ctx.N = dzs, row0, N
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        outs = []
        for d in ctx.dzs:
            if d.shape[0] == ctx.N:
                outs.append(d * g)
            else:  # gathered batch: only the local rows carry gradient
                full = torch.zeros(ctx.N, d.shape[1], dtype=F32, device=d.device)
                full[ctx.row0:ctx.row0 + d.shape[0]] = d * g
                outs.append(full)
        return (None, None, None, None, None) + tuple(outs)


_WS = {}


def _workspace(N, nmod, device):
    key = (N, nmod, str(device))
    ws = _WS.get(key)
    if ws is None:
        ws = torch.empty(ops.infonce_workspace_floats(N, nmod), dtype=F32, device=device)
        _WS.clear()
        _WS[key] = ws
    return ws


def infonce(zs, labels, scale, row0=0, n_local=None):
    if len(zs) < 2:
        raise ValueError("Too less element for calculating the contrastive loss.")
    if not zs[0].is_cuda:
        raise RuntimeError("bioscanclip: tensors must live on the GPU (no CPU compute path)")
    ws = _workspace(zs[0].shape[0], len(zs), zs[0].device)
    return _InfoNCE.apply(labels.to(torch.int64).contiguous(), float(scale), row0, n_local, ws, *[z.to(F32) for z in zs])
