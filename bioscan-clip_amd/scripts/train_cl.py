"""Contrastive pre-training entry point -- drop-in for reference ``scripts/train_cl.py`` on the HIP path.

    python scripts/train_cl.py 'model_config=lora_vit_lora_barcode_bert_ssl' [key=value ...]
    python -m torch.distributed.run --nproc-per-node N scripts/train_cl.py 'model_config=...'     # one rank per GPU

Keeps the reference's structure (train_cl.py:117-252): ``ddp_setup`` -> dataloader -> ``load_clip_model`` -> broadcast ->
AdamW (+ optional one_cycle / exponential / step / cosine scheduler, stepped per iteration) -> ``ContrastiveLoss`` ->
``train_epoch`` per epoch -> ``last.pth`` checkpoint.  Differences, all deliberate:
  * one process per GPU started by torchrun (RANK/LOCAL_RANK/WORLD_SIZE), not ``mp.spawn``; backend "nccl" = RCCL;
  * with world_size > 1 the loss is the all-gathered global-batch loss and the flat trainable gradients are
    all-reduced (the reference never synchronises gradients: SURVEY App. B-1);
  * parameters are broadcast as one flat buffer per encoder (reference: one broadcast per tensor, train_cl.py:29-31);
  * the HDF5 datasets are not available here: ``dataset=synthetic`` (default) feeds synthetic batches of the same layout,
    ``dataset=synthetic_raw`` feeds raw uint8 images + nucleotide strings through the GPU input pipeline (the reference's
    augmentation chain and 5-mer tokeniser as HIP kernels, bioscanclip/util/gpu_pipeline.py);
  * the per-epoch evaluation (``eval_phase``, reference train_cl.py:217-243) runs natively: feature extraction with the HIP
    encoders and top-k retrieval with ``bsclip_topk_ip`` in place of faiss.
"""
import datetime
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
if PKG not in sys.path:
    sys.path.insert(0, PKG)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.optim.lr_scheduler as lr_scheduler  # noqa: E402

from bioscanclip.epoch.train_epoch import train_epoch  # noqa: E402
from bioscanclip.hip import dist as hdist  # noqa: E402
from bioscanclip.hip.optim import FusedAdamW  # noqa: E402
from bioscanclip.model.loss_func import ContrastiveLoss, GlobalBatchContrastiveLoss  # noqa: E402
from bioscanclip.model.simple_clip import load_clip_model  # noqa: E402
from bioscanclip.util.config import load_config  # noqa: E402
from bioscanclip.util.synthetic import SyntheticCLIPLoader, SyntheticEvalLoader, SyntheticRawLoader  # noqa: E402


def print_when_rank_zero(message, rank=0):
    if rank is None or rank == 0:
        print(message)


def broadcast_model(model, rank):
    """Reference train_cl.py:29-31: every parameter from rank 0 (plus buffers).  Random-init / unseeded trunks differ per
    rank otherwise, and the summed LoRA gradients would come from different models."""
    hdist.broadcast_parameters(model, src=0)
    hdist.assert_frozen_in_sync(model)


def eval_phase(model, device, all_keys_dataloader, seen_val_dataloader, unseen_val_dataloader, k_list, args,
               species_to_drop=None, rank=None, for_open_clip=False):
    """Reference train_cl.py:70-83: features of the key / seen / unseen splits on the HIP encoders, then the retrieval
    accuracy table (``bsclip_topk_ip`` in place of faiss)."""
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    from inference_and_eval import get_features_and_label, inference_and_print_result
    keys_dict = get_features_and_label(all_keys_dataloader, model, device, for_key_set=True, for_open_clip=for_open_clip)
    seen_val_dict = get_features_and_label(seen_val_dataloader, model, device, for_open_clip=for_open_clip)
    unseen_val_dict = get_features_and_label(unseen_val_dataloader, model, device, for_open_clip=for_open_clip)
    acc_dict, _, pred_dict = inference_and_print_result(keys_dict, seen_val_dict, unseen_val_dict, args=args,
                                                        small_species_list=None, k_list=k_list)
    return acc_dict, pred_dict


def ddp_setup(rank: int, world_size: int, port):
    """Reference train_cl.py:42-46 (NCCL group + set_device); rendezvous on 127.0.0.1 because container hostnames may
    not resolve."""
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local_rank)
    if world_size > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", rank=rank, world_size=world_size)
    if world_size > 1:
        # host-thread budget for N ranks on one host: the step has no CPU tensor math (three graph launches and a handful of
        # collectives per step, hip/graph.py GraphedDistStep), so each rank keeps a 2-thread intra-op pool -- with the input
        # loader's producer thread and the runtime's helper threads that is <= 6 runnable threads per rank, 48 for 8 ranks
        torch.set_num_threads(int(os.environ.get("BSCLIP_HOST_THREADS", "2")))
    return local_rank


def build_scheduler(args, optimizer, total_steps):
    """Reference train_cl.py:160-181."""
    mc = args.model_config
    if not hasattr(mc, 'lr_scheduler'):
        return None
    if mc.lr_scheduler == 'one_cycle':
        max_lr = 0.001
        if hasattr(mc, 'lr_config') and hasattr(mc.lr_config, 'max_lr'):
            max_lr = mc.lr_config.max_lr
        return lr_scheduler.OneCycleLR(optimizer, max_lr=max_lr, total_steps=total_steps, pct_start=0.3,
                                       anneal_strategy='cos', cycle_momentum=False)
    if mc.lr_scheduler == 'exponential':
        return lr_scheduler.ExponentialLR(optimizer, gamma=0.95)
    if mc.lr_scheduler == 'step':
        return lr_scheduler.StepLR(optimizer, step_size=10, gamma=0.5)
    if mc.lr_scheduler == 'cosine':
        min_lr = 1e-9
        if hasattr(mc, 'lr_config') and hasattr(mc.lr_config, 'min_lr'):
            min_lr = mc.lr_config.min_lr
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=total_steps, eta_min=min_lr)
    return None


def main_process(rank: int, world_size: int, args):
    if getattr(args, "debug_flag", False) or rank != 0:
        args.activate_wandb = False
        args.save_inference = False
        args.save_ckpt = False
    mc = args.model_config
    if not hasattr(mc, "for_open_clip"):
        mc.for_open_clip = False
    local_rank = ddp_setup(rank, world_size, str(getattr(mc, "port", 12316)))
    device = torch.device("cuda", local_rank)

    print_when_rank_zero("Construct dataloader...", rank)
    steps = int(getattr(args, "synthetic_steps_per_epoch", 20))
    with_text = hasattr(mc, 'language')
    dataset = getattr(args, "dataset", "synthetic")
    from bioscanclip.util import shards
    if shards.is_shard(str(dataset)):
        # a pre-decoded shard directory (bioscanclip/util/shards.py): the reference's Dataset_for_CL + prepare() (dataset.py:41-48,
        # 97-275) -- DistributedSampler(drop_last=True) order, pinned double-buffered H2D on a side stream, GPU augmentation and
        # k-mer tokeniser; shuffled like the reference's pre-training loader (train_cl.py: shuffle=True)
        pre_train_dataloader = shards.ShardLoader(str(dataset), int(mc.batch_size), rank=rank, world_size=world_size, shuffle=True,
                                                  seed=int(getattr(args, "seed", 0)), for_training=True, with_text=with_text,
                                                  device=device)
    elif dataset not in ("synthetic", "synthetic_raw"):
        raise NotImplementedError("the HDF5 file itself is outside the accelerated path (SURVEY 8f-3; h5py is absent): convert a "
                                  "split with bioscanclip.util.shards.convert_hdf5_split and pass dataset=<shard directory>, or use "
                                  "dataset=synthetic / dataset=synthetic_raw")
    else:
        Loader = SyntheticRawLoader if dataset == "synthetic_raw" else SyntheticCLIPLoader
        pre_train_dataloader = Loader(int(mc.batch_size), steps, with_text=with_text, rank=rank, world_size=world_size)

    print_when_rank_zero("Initialize model...", rank)
    if not hasattr(args, "allow_random_init"):
        args.allow_random_init = True
    model = load_clip_model(args)
    model = model.to(device)

    total_steps = len(pre_train_dataloader) * mc.epochs
    lr = 0.001
    if hasattr(mc, 'lr_config') and hasattr(mc.lr_config, 'lr'):
        lr = mc.lr_config.lr
    optimizer = FusedAdamW(model.parameters(), lr=lr)
    if world_size > 1 and getattr(mc, "disable_lora", False):
        # full fine-tuning on several ranks: each rank keeps the AdamW moments of one slice of every flat buffer (3.2 GB / W for
        # I+D+T instead of 3.2 GB per rank) and broadcasts its updated slice (hip/optim.py shard_state)
        optimizer.shard_state()
    scheduler = build_scheduler(args, optimizer, total_steps)

    if mc.for_open_clip:
        raise NotImplementedError("the open_clip branch is not part of the HIP-accelerated path")
    if world_size > 1:
        criterion = GlobalBatchContrastiveLoss(criterion=nn.CrossEntropyLoss(), logit_scale=1 / 0.07)
    else:
        criterion = ContrastiveLoss(criterion=nn.CrossEntropyLoss(), logit_scale=1 / 0.07)

    print_when_rank_zero("training...", rank)
    folder_path = None
    if getattr(args, "save_ckpt", False):
        stamp = datetime.datetime.now().strftime("%Y-%m-%d_%H%M%S")
        folder_path = os.path.join(getattr(args, "project_root_path", "."), getattr(args, "model_output_dir", "ckpt"),
                                   str(getattr(mc, "model_output_name", "bioscan_clip_hip")), stamp)
        os.makedirs(folder_path, exist_ok=True)

    broadcast_model(model, rank)  # before the first step, like the reference (train_cl.py:149)
    eval_loaders = None
    if getattr(args, "synthetic_eval", False):
        n_eval = int(getattr(args, "synthetic_eval_batches", 2))
        eval_loaders = [SyntheticEvalLoader(min(int(mc.batch_size), 24), n_eval, with_text=with_text, seed=s_)
                        for s_ in (7001, 7002, 7003)]  # keys, seen, unseen
    k_list = [1, 3, 5]
    best_epoch, best_overall_acc = None, None
    losses = []
    for epoch in range(mc.epochs):
        if hasattr(pre_train_dataloader, "set_epoch"):
            pre_train_dataloader.set_epoch(epoch)   # a new shuffle per epoch (DistributedSampler.set_epoch)
        losses.append(train_epoch(getattr(args, "activate_wandb", False), mc.epochs, epoch, pre_train_dataloader, model,
                                  optimizer, criterion, device, rank=rank, scheduler=scheduler,
                                  for_open_clip=False))
        period = int(getattr(mc, "evaluation_period", 1))
        if (epoch % period == 0 or epoch == mc.epochs - 1) and rank == 0:
            if folder_path is not None:
                last_ckpt_path = os.path.join(folder_path, 'last.pth')
                torch.save(model.state_dict(), last_ckpt_path)  # reference key names: loadable by the reference
                print(f'Last ckpt: {last_ckpt_path}')
            if eval_loaders is not None:
                acc_dict, _ = eval_phase(model, device, *eval_loaders, k_list, args=args, rank=rank)
                cell = acc_dict['encoded_image_feature']['encoded_image_feature']
                overall_acc = (cell['seen']['micro_acc'][1]['species'] + cell['unseen']['micro_acc'][1]['species']) / 2
                if best_overall_acc is None or best_overall_acc < overall_acc:
                    best_epoch, best_overall_acc = epoch, overall_acc
                    if folder_path is not None:
                        torch.save(model.state_dict(), os.path.join(folder_path, 'best.pth'))
                print(f'epoch {epoch}: overall_acc {overall_acc:.4f} (best {best_overall_acc:.4f} @ {best_epoch})')
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()
    return losses


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    config_dir = os.path.join(PKG, "bioscanclip", "config")
    args = load_config(config_dir, argv)
    if "model_config" not in args:
        raise SystemExit("usage: train_cl.py 'model_config=<name>' [key=value ...]")
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    print_when_rank_zero(f'world_size: {world_size}', rank)
    return main_process(rank, world_size, args)


if __name__ == '__main__':
    main()
