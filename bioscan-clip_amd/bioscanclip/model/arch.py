"""Parameter containers for the three frozen backbones.

The reference obtains its backbones from third-party packages:
  * timm ``vit_base_patch16_224``           (reference ``bioscanclip/model/simple_clip.py:150``)
  * HF ``BertForMaskedLM(BertConfig(vocab_size=1027))``  (``bioscanclip/model/dna_encoder.py:14-22``)
  * HF ``BertModel`` "prajjwal1/bert-small" (``bioscanclip/model/language_encoder.py:12-20``)

Neither timm nor pretrained weights exist in this image, and this repo computes
nothing with torch modules anyway: all arithmetic runs in the HIP library.  What the
LoRA wrappers (``LoRA_ViT_timm`` etc.) need from a backbone is only its *parameter
tree* -- attribute paths and ``state_dict`` key names identical to the third-party
classes, so reference checkpoints load unchanged.  The classes
below provide exactly that tree.  They deliberately have no ``forward`` arithmetic:
calling one raises, because the only compute path is the HIP engine reached through
the LoRA wrapper modules.

A real timm ``VisionTransformer`` / HF ``BertModel`` instance can be passed to the
wrappers instead of these containers; the wrappers only touch the attribute paths
that both share.
"""
import math

import torch
import torch.nn as nn


class _NoForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard only
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container; arithmetic runs in the HIP engine "
            "through LoRA_ViT_timm / LoRA_barcode_bert / LoRA_bert (no torch fallback exists)."
        )


# --------------------------------------------------------------------------------------
# timm-0.6.13 VisionTransformer parameter tree
# --------------------------------------------------------------------------------------
class _PatchEmbed(_NoForward):
    def __init__(self, in_chans, embed_dim, patch):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch, stride=patch)


class _VitAttention(_NoForward):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _VitMlp(_NoForward):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _VitBlock(_NoForward):
    def __init__(self, dim, num_heads, mlp_ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _VitAttention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _VitMlp(dim, int(dim * mlp_ratio))


class VisionTransformerParams(_NoForward):
    """Same parameter names as timm ``vit_base_patch16_224`` (App. A.1):
    ``cls_token, pos_embed, patch_embed.proj.*, blocks.{i}.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}.*,
    norm.*, head.*``."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.num_classes = num_classes
        n_patches = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, n_patches + 1, embed_dim) * 0.02)
        self.patch_embed = _PatchEmbed(in_chans, embed_dim, patch_size)
        self.blocks = nn.Sequential(*[_VitBlock(embed_dim, num_heads, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        # timm init_weights(''): trunc_normal_(std=.02) on linears, zeros on biases, normal cls
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def reset_classifier(self, num_classes, global_pool=None):
        """timm ``VisionTransformer.reset_classifier``: fresh ``head`` with torch's default Linear init
        (called from reference ``image_encoder.py:94-95``)."""
        self.num_classes = num_classes
        self.head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()


def vit_base_patch16_224(**kw):
    return VisionTransformerParams(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, **kw)


# --------------------------------------------------------------------------------------
# HF BertModel / BertForMaskedLM parameter tree
# --------------------------------------------------------------------------------------
class BertConfigLite:
    """The subset of ``transformers.BertConfig`` this path depends on (defaults = HF defaults)."""

    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                 max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, **_ignored):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        self.max_position_embeddings = max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.layer_norm_eps = layer_norm_eps


class _BertEmbeddings(_NoForward):
    def __init__(self, c):
        super().__init__()
        # HF BertEmbeddings: padding_idx = config.pad_token_id (default 0): that row never receives a gradient
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class _BertSelfAttention(_NoForward):
    def __init__(self, c):
        super().__init__()
        self.num_attention_heads = c.num_attention_heads
        self.query = nn.Linear(c.hidden_size, c.hidden_size)
        self.key = nn.Linear(c.hidden_size, c.hidden_size)
        self.value = nn.Linear(c.hidden_size, c.hidden_size)


class _BertSelfOutput(_NoForward):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class _BertAttention(_NoForward):
    def __init__(self, c):
        super().__init__()
        self.self = _BertSelfAttention(c)
        self.output = _BertSelfOutput(c)


class _BertIntermediate(_NoForward):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.intermediate_size)


class _BertOutput(_NoForward):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.intermediate_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class _BertLayer(_NoForward):
    def __init__(self, c):
        super().__init__()
        self.attention = _BertAttention(c)
        self.intermediate = _BertIntermediate(c)
        self.output = _BertOutput(c)


class _BertEncoder(_NoForward):
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList([_BertLayer(c) for _ in range(c.num_hidden_layers)])


class _BertPooler(_NoForward):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)


def _hf_bert_init(module, std=0.02):
    """HF ``BertPreTrainedModel._init_weights``: N(0, 0.02) for Linear/Embedding, LN = (1, 0)."""
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, mean=0.0, std=std)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Embedding):
            nn.init.normal_(m.weight, mean=0.0, std=std)
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)


class BertModelParams(_NoForward):
    """Parameter tree of HF ``BertModel`` (keys ``embeddings.*, encoder.layer.{i}.*, pooler.dense.*``)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = _BertEmbeddings(config)
        self.encoder = _BertEncoder(config)
        self.pooler = _BertPooler(config)
        _hf_bert_init(self)


class _BertPredictionHeadTransform(_NoForward):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class _BertLMPredictionHead(_NoForward):
    def __init__(self, c):
        super().__init__()
        self.transform = _BertPredictionHeadTransform(c)
        self.decoder = nn.Linear(c.hidden_size, c.vocab_size)
        # HF keeps the tied output bias as a separate key; after the reference swaps ``decoder`` for a
        # fresh Linear(768, num_classes) (dna_encoder.py:93-95) this [vocab] vector stays in the
        # state_dict, unused.
        self.bias = nn.Parameter(torch.zeros(c.vocab_size))


class _BertOnlyMLMHead(_NoForward):
    def __init__(self, c):
        super().__init__()
        self.predictions = _BertLMPredictionHead(c)


class _BertNoPooler(_NoForward):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = _BertEmbeddings(config)
        self.encoder = _BertEncoder(config)


class BertForMaskedLMParams(_NoForward):
    """Parameter tree of HF ``BertForMaskedLM`` (``bert.*`` without pooler, ``cls.predictions.*``)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.bert = _BertNoPooler(config)
        self.cls = _BertOnlyMLMHead(config)
        _hf_bert_init(self)


def barcode_bert_config(k=5, **overrides):
    """``BertConfig(vocab_size=len(vocab))`` with vocab = 3 specials + 4**k k-mers (dna_encoder.py:14-19)."""
    return BertConfigLite(vocab_size=4 ** k + 3, **overrides)


def bert_small_config(**overrides):
    """prajjwal1/bert-small: hidden 512, 4 layers, 8 heads, FFN 2048 (public model card)."""
    kw = dict(vocab_size=30522, hidden_size=512, num_hidden_layers=4, num_attention_heads=8,
              intermediate_size=2048)
    kw.update(overrides)
    return BertConfigLite(**kw)


def lora_init_(w_a: nn.Linear, w_b: nn.Linear):
    """Reference ``reset_parameters`` (image_encoder.py:102-106): A ~ kaiming_uniform(a=sqrt 5), B = 0."""
    nn.init.kaiming_uniform_(w_a.weight, a=math.sqrt(5))
    nn.init.zeros_(w_b.weight)
