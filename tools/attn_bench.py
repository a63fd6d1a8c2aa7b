"""Times bsclip_attn_fwd / bsclip_attn_bwd at the step's shapes (B=256: ViT S=197 no dropout, DNA S=133 dropout 0.1): the backward
re-hashing its dropout masks and reading the forward's keep-bit words (round 5), the forward with and without writing them.
BSCLIP_ATTN_PRELOAD=0 selects the no-dropout backward without the preloaded row constant (A/B: run the tool twice).
LAYOUT=head times the same kernels on head-major copies of the operands ((batch, head) items contiguous: [B*heads, S, 192] / [.., 64])."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch  # noqa: E402

from bioscanclip.hip import ops  # noqa: E402

B = int(os.environ.get("B", "256"))
HEAD = os.environ.get("LAYOUT", "token") == "head"
print(f"attn_bench: B={B} layout={'head-major' if HEAD else 'token-major'} BSCLIP_ATTN_PRELOAD={os.environ.get('BSCLIP_ATTN_PRELOAD', '1')}")
for name, S, p in (("vit", 197, 0.0), ("dna", 133, 0.1), ("dna-nodrop", 133, 0.0)):
    heads, H = 12, 768
    if HEAD:   # one head per "batch entry": the same kernels, every item contiguous
        Bk, hk = B * heads, 1
    else:
        Bk, hk = B, heads
    Hk = hk * 64
    qkv = (torch.randn(Bk * S, 3 * Hk, device="cuda") * 0.5).bfloat16()
    dctx = torch.randn(Bk * S, Hk, device="cuda").bfloat16()
    ctx = torch.empty(Bk * S, Hk, device="cuda", dtype=torch.bfloat16)
    dqkv = torch.empty(Bk * S, 3 * Hk, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(Bk, hk, S, device="cuda")
    drop = (p, 1234) if p else None
    bits = torch.zeros(Bk * hk * S * ops.KEEP_WORDS, device="cuda", dtype=torch.int32) if p else None
    res = {}
    cases = [("fwd", lambda: ops.attn_fwd(qkv, Bk, S, hk, 0.125, ctx, lse, dropout=drop)),
             ("bwd", lambda: ops.attn_bwd(qkv, dctx, lse, Bk, S, hk, 0.125, dqkv, dropout=drop))]
    if p:
        cases += [("fwd+bits", lambda: ops.attn_fwd(qkv, Bk, S, hk, 0.125, ctx, lse, dropout=drop, keep_bits=bits)),
                  ("bwd+bits", lambda: ops.attn_bwd(qkv, dctx, lse, Bk, S, hk, 0.125, dqkv, dropout=drop, keep_bits=bits))]
    if not HEAD:   # the LoRA partial sums out of the backward (round 5) and the two reductions that follow / the pass they replace
        M = B * S
        h = torch.randn(M, H + 64, device="cuda").bfloat16()
        lb = torch.randn(2, H, 4, device="cuda") * 0.1
        dtp, dbp = torch.empty(heads, 2, M, 4, device="cuda"), torch.empty(B * heads, 2, 4, 64, device="cuda")
        dt, dA = torch.empty(M, 8, device="cuda"), torch.zeros(8, H, device="cuda")
        dBq, dBv = torch.zeros(H, 4, device="cuda"), torch.zeros(H, 4, device="cuda")
        cases += [("bwd+lora", lambda: ops.attn_bwd(qkv, dctx, lse, B, S, heads, 0.125, dqkv, dropout=drop, keep_bits=bits,
                                                    lora=(h[:, H:], lb, dtp, dbp))),
                  ("lora_grad", lambda: ops.lora_grad(dqkv, h, M, H, lb, dt, dA, dBq, dBv)),
                  ("lora_heads", lambda: ops.lora_grad_heads(h, M, H, B, dtp, dbp, dt, dA, dBq, dBv))]
    for what, fn in cases:
        fn()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        res[what] = best
    fl = 4.0 * B * heads * S * S * 64
    line = (f"{name:11s} S={S} p={p}: fwd {res['fwd']*1e3:7.1f} us ({fl/res['fwd']/1e9:6.1f} TF)   "
            f"bwd {res['bwd']*1e3:7.1f} us ({2.5*fl/res['bwd']/1e9:6.1f} TF at 5 products)")
    if p:
        line += f"   |  keep-bit words: fwd {res['fwd+bits']*1e3:7.1f} us   bwd {res['bwd+bits']*1e3:7.1f} us"
    if "bwd+lora" in res:
        base = res["bwd+bits"] if p else res["bwd"]
        line += (f"\n{'':11s} LoRA partials: bwd {res['bwd+lora']*1e3:7.1f} us (+{(res['bwd+lora']-base)*1e3:5.1f})   lora_grad_heads {res['lora_heads']*1e3:6.1f} us"
                 f"   vs lora_grad {res['lora_grad']*1e3:6.1f} us   net {((res['bwd+lora']-base)+res['lora_heads']-res['lora_grad'])*1e3:+6.1f} us per layer")
    print(line, flush=True)
