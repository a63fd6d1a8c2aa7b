"""SimpleCLIP and ``load_clip_model`` -- drop-in for reference ``bioscanclip/model/simple_clip.py``.

``SimpleCLIP.forward`` keeps the reference's order and output convention (simple_clip.py:27-50): DNA, image and
text encoders, each followed by ``F.normalize(p=2, dim=-1)`` (here the HIP l2norm kernel), ``None`` for an absent
modality.  The open_clip branch (simple_clip.py:36-44,141-145) is outside the accelerated path
and raises.
"""
import torch
import torch.nn as nn

from bioscanclip.hip.dist import start_gather
from bioscanclip.hip.engine import forked_from
from bioscanclip.hip.functional import l2_normalize
from bioscanclip.model.arch import vit_base_patch16_224
from bioscanclip.model.dna_encoder import Freeze_DNA_Encoder, LoRA_barcode_bert, load_pre_trained_bioscan_bert
from bioscanclip.model.image_encoder import LoRA_ViT_timm
from bioscanclip.model.language_encoder import LoRA_bert, load_pre_trained_bert


_TOWER_STREAMS = __import__("os").environ.get("BSCLIP_TOWER_STREAMS", "1") != "0"
# host enqueue order of the towers: the longer (image) tower first measured 45.40 vs 45.72 ms/step with DNA first
_IMAGE_FIRST = __import__("os").environ.get("BSCLIP_IMAGE_FIRST", "1") == "1"
# debugging: tower indices (0 = DNA, 1 = image, 2 = text) that stay on the caller's stream while the others fork, e.g. "2"
_SERIAL_TOWERS = {int(i) for i in __import__("os").environ.get("BSCLIP_TOWER_SERIAL", "").split(",") if i.strip()}
_streams = {}


def _tower_stream(k, device):
    key = (k, device.index)
    if key not in _streams:
        _streams[key] = torch.cuda.Stream(device=device)
    return _streams[key]


class SimpleCLIP(nn.Module):
    def __init__(self, image_encoder, dna_encoder, language_encoder, open_clip_model=None):
        super(SimpleCLIP, self).__init__()
        if open_clip_model is not None:
            raise NotImplementedError("the open_clip (ViT-L/14) branch is not part of the HIP-accelerated path")
        self.image_encoder = image_encoder
        self.dna_encoder = dna_encoder
        self.language_encoder = language_encoder
        self.open_clip_model = None
        self.tokenizer_for_open_clip = None

    def forward(self, image_input, dna_input, language_input):
        image_output = None
        dna_output = None
        language_output = None

        # The towers are independent until the loss: each runs on its own HIP stream (forward here, backward through
        # autograd, which replays a node on the stream of its forward), so one tower's memory-bound kernels and
        # tail waves fill the other's idle CUs.  Disable with BSCLIP_TOWER_STREAMS=0.
        towers = [(self.dna_encoder, dna_input), (self.image_encoder, image_input),
                  (self.language_encoder, language_input)]
        outs = [None, None, None]
        use_streams = _TOWER_STREAMS and sum(enc is not None for enc, _ in towers) > 1 and torch.cuda.is_available()
        cur = torch.cuda.current_stream() if use_streams else None
        order = (1, 0, 2) if _IMAGE_FIRST else (0, 1, 2)
        for k in order:
            enc, x = towers[k]
            if enc is None:
                continue
            if use_streams and k not in _SERIAL_TOWERS and not isinstance(enc, Freeze_DNA_Encoder):
                side = _tower_stream(k, cur.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side), forked_from(cur):
                    y = l2_normalize(enc(x))
                if not torch.cuda.is_current_stream_capturing():   # a capture's private pool owns the tensor's lifetime
                    y.record_stream(cur)
                outs[k] = (y, side)
            else:
                outs[k] = (l2_normalize(enc(x)), None)
        # global-batch loss: every modality's all-gather starts here, each ordered behind ITS tower's stream only -- and issued
        # shortest tower first (text, DNA, image): the process group runs its collectives on one stream in issue order, so a gather
        # issued behind the image tower's would wait for the image tower (tools/dist_overlap_probe.py)
        for k in (2, 0, 1):
            if outs[k] is not None:
                y, side = outs[k]
                if side is not None:
                    with torch.cuda.stream(side):
                        start_gather(y)
                else:
                    start_gather(y)
        for k in range(3):
            if outs[k] is not None:
                y, side = outs[k]
                if side is not None:
                    cur.wait_stream(side)
                outs[k] = y
        dna_output, image_output, language_output = outs
        return image_output, dna_output, language_output


def _load_vit(args):
    """timm.create_model('vit_base_patch16_224', pretrained=True) (simple_clip.py:150) needs timm + a download.
    Here: the same parameter tree; weights from ``args.vit_checkpoint`` (a timm state_dict file) when given,
    otherwise timm's random init (synthetic benchmarks)."""
    vit = vit_base_patch16_224()
    ckpt = getattr(args, "vit_checkpoint", None) if args is not None else None
    if ckpt:
        from bioscanclip.util.util import load_checked
        # timm checkpoints carry the 1000-class ImageNet head; the head is replaced by reset_classifier right after
        load_checked(vit, torch.load(ckpt, map_location="cpu"), f"ViT checkpoint {ckpt}", allow_missing=("head.",),
                     allow_unexpected=("head.", "fc_norm."))
    return vit


def load_clip_model(args, device=None):
    """Reference simple_clip.py:125-203, same config keys (``args.model_config.{image,dna,language,output_dim,
    disable_lora}``, ``args.bioscan_bert_checkpoint``).  ``disable_lora: true`` selects full fine-tuning (hip/engine_ft.py).
    MLP / open_clip variants are outside the accelerated path and raise instead of silently running something else."""
    image_encoder = None
    dna_encoder = None
    language_encoder = None
    mc = args.model_config

    disable_lora = False
    if hasattr(mc, 'disable_lora'):
        disable_lora = mc.disable_lora
    # disable_lora: the reference passes lora_layer=[] to every wrapper and unfreezes all parameters afterwards
    # (simple_clip.py:151-153,165-167,182-184,199-201).  [] means "no LoRA" for the BERT wrappers and -- being falsy -- "LoRA on
    # every block" for the ViT wrapper; both behaviours are kept.
    ll = [] if disable_lora else None
    if mc.image.model == "lora_clip_image" and hasattr(mc, 'language') and mc.language.model == "lora_clip_text":
        raise NotImplementedError("the open_clip (ViT-L/14) branch is not part of the HIP-accelerated path")

    if mc.image.input_type == "image":
        image_encoder = LoRA_ViT_timm(vit_model=_load_vit(args), r=4, num_classes=mc.output_dim, lora_layer=ll)
    else:
        raise NotImplementedError("feature-input MLP encoders are outside the accelerated path")

    if hasattr(mc, 'language'):
        if mc.language.input_type == "sequence":
            _, pre_trained_bert = load_pre_trained_bert(getattr(args, "bert_small_checkpoint", None))
            language_encoder = LoRA_bert(model=pre_trained_bert, r=4, num_classes=mc.output_dim, lora_layer=ll)
        else:
            raise TypeError(f"Using {mc.language.input_type} as language input is not support yet.")

    if hasattr(mc, 'dna'):
        if hasattr(mc.dna, 'freeze') and mc.dna.freeze:
            dna_encoder = Freeze_DNA_Encoder()
        elif mc.dna.input_type == "sequence":
            if mc.dna.model == "lora_barcode_bert":
                ckpt = getattr(args, "bioscan_bert_checkpoint", None)
                if ckpt is not None and not __import__("os").path.exists(str(ckpt)):
                    if not getattr(args, "allow_random_init", False):
                        raise FileNotFoundError(f"BarcodeBERT checkpoint {ckpt} not found "
                                                "(set allow_random_init=true for synthetic runs)")
                    ckpt = None
                pre_trained_barcode_bert = load_pre_trained_bioscan_bert(bioscan_bert_checkpoint=ckpt)
                dna_encoder = LoRA_barcode_bert(model=pre_trained_barcode_bert, r=4, num_classes=mc.output_dim, lora_layer=ll)
        else:
            raise NotImplementedError("feature-input MLP encoders are outside the accelerated path")

    model = SimpleCLIP(image_encoder=image_encoder, dna_encoder=dna_encoder, language_encoder=language_encoder)

    if device is not None:
        model.to(device)

    if disable_lora:
        enable_full_fine_tuning(model)

    return model


def enable_full_fine_tuning(model):
    """Reference simple_clip.py:199-201: ``for param in model.parameters(): param.requires_grad = True``; the encoders then build
    their full fine-tuning engines (hip/engine_ft.py) at the next forward."""
    for param in model.parameters():
        param.requires_grad = True
    for enc in (model.image_encoder, model.dna_encoder, model.language_encoder):
        if enc is not None and not isinstance(enc, Freeze_DNA_Encoder):
            enc.hip_full_ft = True
            enc._engine = None
    return model
