"""Training epoch driver with the reference's signature (``bioscanclip/epoch/train_epoch.py:11-61``).

Per batch, in the reference's order: inputs to the device, ``optimizer.zero_grad()``, ``model(image, dna, text)``,
``criterion(...)``, ``backward()``, ``optimizer.step()``, ``scheduler.step()``.  What differs, deliberately:
autograd anomaly mode stays off, the loss is read back once per step (the reference synchronises three times), tqdm / wandb
are optional, and under a process group of more than one rank the flat trainable gradients are all-reduced (SUM) before the
optimizer step.  Returns the mean loss of the epoch.
"""
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
    for step, batch in steps:
        image, dna, text, label = _to_device(batch, device)
        optimizer.zero_grad()
        if hasattr(criterion, "prefetch_labels"):
            criterion.prefetch_labels(label)  # global-batch loss: the label all-gather starts before the encoders
        loss = criterion(*model(image, dna, text), label)
        loss.backward()
        hdist.allreduce_grads(model)  # no-op without a multi-rank process group
        if hasattr(optimizer, "needs_attach") and optimizer.needs_attach():
            optimizer.attach(model)  # FusedAdamW: adopt the engines' flat buffers once they exist (or were rebuilt)
        optimizer.step()
        if scheduler is not None:
            scheduler.step()

        value = loss.item()  # the step's only host synchronisation
        running += value
        lr = optimizer.param_groups[0]["lr"]
        if show:
            mem = f" || Allocated: {torch.cuda.memory_allocated() / 2 ** 30:.2f} GB" if check_cuda_memory else ""
            steps.set_description(f"Epoch: {epoch}||Step: {step}/{n_steps}||Loss: {value}{mem} || Current LR: {lr}")
        if activate_wandb and wandb is not None:
            wandb.log({"loss": value, "step": step + epoch * n_steps, "learning_rate": lr})

    mean_loss = running / max(n_steps, 1)
    print(f"Epoch [{epoch}/{total_epochs}], Loss: {mean_loss}")
    return mean_loss
