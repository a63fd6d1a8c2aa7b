"""Training epoch driver with the reference's signature (``bioscanclip/epoch/train_epoch.py:11-61``).

Per batch, in the reference's order: inputs to the device, ``optimizer.zero_grad()``, ``model(image, dna, text)``,
``criterion(...)``, ``backward()``, ``optimizer.step()``, ``scheduler.step()``.  What differs, deliberately:
autograd anomaly mode (switched on by the reference in every epoch, ``train_epoch.py:12``) is off unless asked for with
``BSCLIP_DETECT_ANOMALY=1`` / ``hip.engine.set_detect_anomaly(True)`` -- then every tower's output and gradients and the loss are
checked for Inf / NaN on the device and a RuntimeError names where they first appear, on the eager launch path --, the loss is
read back once per step (the reference synchronises three times), tqdm / wandb
are optional, and under a process group of more than one rank the flat trainable gradients are all-reduced (SUM) before the
optimizer step.  Returns the mean loss of the epoch.

Launch path (round 3): on one rank the step runs as ONE replayed hipGraph (``bioscanclip.hip.graph.GraphedStep``: the same
Python body captured once -- bitwise the eager step, tests/test_30_graph_gpu.py) and the loss is read one step late, so the host
never waits for the step it has just enqueued: what ``bench.py`` measures is what ``scripts/train_cl.py`` runs.  A batch whose
shape differs from the captured one (a last, smaller batch) is enqueued eagerly.  ``BSCLIP_GRAPH=0`` forces the eager loop.  With
more than one rank the step is a set of per-tower captured graphs (forward_k | loss | backward_k | AdamW) with each tower's
all-gather and all-reduce issued eagerly from that tower's stream between them (``GraphedDistStep``).
"""
import os
import time

import torch

try:  # neither ships in this image
    from tqdm import tqdm
except Exception:  # pragma: no cover
    tqdm = None
try:
    import wandb
except Exception:  # pragma: no cover
    wandb = None


def _to_device(batch, device):
    """The reference's 7-tuple ``(processid, image, dna, input_ids, token_type_ids, attention_mask, label)`` -> model inputs."""
    _, image, dna, input_ids, token_type_ids, attention_mask, label = batch
    text = None
    if input_ids is not None:
        text = {"input_ids": input_ids.to(device), "token_type_ids": token_type_ids.to(device),
                "attention_mask": attention_mask.to(device)}
    image = None if image is None else image.to(device)
    dna = None if dna is None else dna.to(device)
    return image, dna, text, label.to(device)


def _graphed_step(model, optimizer, criterion, device):
    """The captured step for this (model, optimizer, criterion), created on first use and kept on the model; None when the
    step has to stay eager: BSCLIP_GRAPH=0, a CPU device, an optimizer other than FusedAdamW, more than one rank."""
    from bioscanclip.hip import dist as hdist
    from bioscanclip.hip import engine
    if os.environ.get("BSCLIP_GRAPH", "1") == "0" or torch.device(device).type != "cuda" or engine.DETECT_ANOMALY:
        return None
    if not hasattr(optimizer, "enable_device_hyper"):
        return None
    multi = not hdist._inactive(None)
    if multi != hasattr(criterion, "prefetch_labels"):     # a process group without the global-batch loss (or the reverse): eager
        return None
    from bioscanclip.hip.graph import GraphedDistStep, GraphedStep
    key = (id(optimizer), id(criterion))
    g = getattr(model, "_bsclip_graphed", None)
    if g is None or g[0] != key:
        # more than one rank: per-tower captured graphs with the collectives issued eagerly between them (GraphedDistStep)
        g = (key, (GraphedDistStep if multi else GraphedStep)(model, optimizer, criterion, warmup=2))
        model._bsclip_graphed = g
    return g[1]


def train_epoch(activate_wandb, total_epochs, epoch, dataloader, model, optimizer, criterion, device, scheduler=None,
                for_open_clip=False, rank=None, check_cuda_memory=False):
    from bioscanclip.hip import dist as hdist
    if for_open_clip:
        raise NotImplementedError("the open_clip branch is not part of the HIP-accelerated path")
    n_steps = len(dataloader)
    show = rank == 0 and tqdm is not None
    steps = tqdm(enumerate(dataloader), total=n_steps) if show else enumerate(dataloader)
    running = 0.0

    model.train()
    graphed = _graphed_step(model, optimizer, criterion, device)
    ring = torch.zeros(4, dtype=torch.float32, device=device) if graphed is not None else None
    # one HIP event per ring slot: the host sleeps between hipEventQuery polls until the step it wants to read has finished.
    # Every synchronising call of this runtime spin-waits (``.item()``, hipEventSynchronize -- blocking-sync events included:
    # bench.py measured 100 % of a core either way): a core per rank burnt for the whole epoch, eight of them on an 8-GPU node.
    done = [torch.cuda.Event() for _ in range(4)] if graphed is not None else None
    pending = None   # (step, lr) whose loss sits in ring[step % 4] and has not been read yet

    def read(slot):
        while not done[slot].query():
            time.sleep(5e-4)
        return ring[slot].item()

    def report(step, value, lr):
        if show:
            mem = f" || Allocated: {torch.cuda.memory_allocated() / 2 ** 30:.2f} GB" if check_cuda_memory else ""
            steps.set_description(f"Epoch: {epoch}||Step: {step}/{n_steps}||Loss: {value}{mem} || Current LR: {lr}")
        if activate_wandb and wandb is not None:
            wandb.log({"loss": value, "step": step + epoch * n_steps, "learning_rate": lr})

    for step, batch in steps:
        image, dna, text, label = _to_device(batch, device)
        if graphed is not None and graphed.accepts(image, dna, text, label):
            loss = graphed(image, dna, text, label)              # device scalar, overwritten by the next replay ...
            ring[step % 4].copy_(loss.detach())                  # ... so it is parked in this step's slot
            done[step % 4].record()
            lr = optimizer.param_groups[0]["lr"]
            if scheduler is not None:
                scheduler.step()
            if pending is not None:                              # read the PREVIOUS step's loss: the host stays one step ahead
                value = read(pending[0] % 4)
                running += value
                report(pending[0], value, pending[1])
            pending = (step, lr)
            continue
        optimizer.zero_grad()
        if hasattr(criterion, "prefetch_labels"):
            criterion.prefetch_labels(label)  # global-batch loss: the label all-gather starts before the encoders
        loss = criterion(*model(image, dna, text), label)
        loss.backward()
        hdist.allreduce_grads(model)  # no-op without a multi-rank process group
        if hasattr(optimizer, "needs_attach") and optimizer.needs_attach():
            optimizer.attach(model)  # FusedAdamW: adopt the engines' flat buffers once they exist (or were rebuilt)
        optimizer.step()
        lr = optimizer.param_groups[0]["lr"]
        if scheduler is not None:
            scheduler.step()
        if pending is not None:
            value = read(pending[0] % 4)
            running += value
            report(pending[0], value, pending[1])
            pending = None
        value = loss.item()  # the eager step's only host synchronisation
        if value != value or value in (float("inf"), float("-inf")):
            from bioscanclip.hip import engine
            if engine.DETECT_ANOMALY:
                raise RuntimeError(f"non-finite loss {value} at step {step} of epoch {epoch} (BSCLIP_DETECT_ANOMALY)")
        running += value
        report(step, value, lr)
    if pending is not None:
        value = read(pending[0] % 4)
        running += value
        report(pending[0], value, pending[1])

    mean_loss = running / max(n_steps, 1)
    print(f"Epoch [{epoch}/{total_epochs}], Loss: {mean_loss}")
    return mean_loss
