"""Training epoch driver -- drop-in for reference ``bioscanclip/epoch/train_epoch.py:11-61`` (same signature and
loop body: H2D, zero_grad, forward, loss, backward, optimizer.step, scheduler.step per iteration).

Differences, all deliberate: autograd anomaly mode is not switched on; ``loss.item()`` is read once
per step (the reference syncs three times); wandb/tqdm are optional; with a process group of world_size > 1 the
flat trainable gradients are all-reduced (SUM) before the optimizer step.
"""
import torch

try:  # optional, as in SURVEY 5 (neither ships in this image)
    from tqdm import tqdm
except Exception:  # pragma: no cover
    tqdm = None
try:
    import wandb
except Exception:  # pragma: no cover
    wandb = None


def train_epoch(activate_wandb, total_epochs, epoch, dataloader, model, optimizer, criterion, device, scheduler=None,
                for_open_clip=False, rank=None, check_cuda_memory=False):
    from bioscanclip.hip import dist as hdist
    if for_open_clip:
        raise NotImplementedError("the open_clip branch is not part of the HIP-accelerated path")
    if rank == 0 and tqdm is not None:
        pbar = tqdm(enumerate(dataloader), total=len(dataloader))
    else:
        pbar = enumerate(dataloader)
    epoch_loss = 0.0
    total_step = len(dataloader)

    model.train()
    for step, batch in pbar:
        processid_batch, image_input_batch, dna_input_batch, input_ids, token_type_ids, attention_mask, label_for_train_batch = batch
        language_input = None
        if input_ids is not None:
            language_input = {'input_ids': input_ids.to(device), 'token_type_ids': token_type_ids.to(device),
                              'attention_mask': attention_mask.to(device)}
        optimizer.zero_grad()
        image_input_batch = image_input_batch.to(device) if image_input_batch is not None else None
        dna_input_batch = dna_input_batch.to(device) if dna_input_batch is not None else None
        image_output, dna_output, language_output = model(image_input_batch, dna_input_batch, language_input)

        label_for_train_batch = label_for_train_batch.to(device)

        loss = criterion(image_output, dna_output, language_output, label_for_train_batch)
        loss.backward()
        hdist.allreduce_grads(model)

        if hasattr(optimizer, "attach") and not getattr(optimizer, "_flats", None):
            optimizer.attach(model)
        optimizer.step()
        if scheduler is not None:
            scheduler.step()

        loss_value = loss.item()
        epoch_loss = epoch_loss + loss_value
        current_lr = optimizer.param_groups[0]['lr']

        if rank == 0 and tqdm is not None:
            mem = ""
            if check_cuda_memory:
                mem = f" || Allocated: {torch.cuda.memory_allocated() / (1024 ** 3):.2f} GB"
            pbar.set_description(f'Epoch: {epoch}||Step: {step}/{total_step}||Loss: {loss_value}{mem} || Current LR: {current_lr}')

        if activate_wandb and wandb is not None:
            wandb.log({"loss": loss_value, "step": step + epoch * len(dataloader), "learning_rate": current_lr})

    print(f'Epoch [{epoch}/{total_epochs}], Loss: {epoch_loss / max(len(dataloader), 1)}')
    return epoch_loss / max(len(dataloader), 1)
