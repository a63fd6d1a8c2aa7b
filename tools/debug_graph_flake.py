"""Debug: tests/test_30_graph_gpu.py::test_graph_replay_equals_eager_step[True-False-True] (exact mode, three towers) repeated in one
process with per-step snapshots of every trainable parameter: at the first mismatch, which step and which tensors diverge first."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bioscan-clip_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

from bioscanclip.hip import engine  # noqa: E402
from oracle import synth  # noqa: E402
from test_30_graph_gpu import _build  # noqa: E402

EXACT = os.environ.get("EXACT", "1") == "1"
if EXACT:
    engine.RESID_STREAM_BF16 = False
    engine.GRAD_STREAM_BF16 = False
    engine.EXACT_FORWARD = True
from bioscanclip.hip.graph import GraphedStep  # noqa: E402
from bioscanclip.hip.optim import FusedAdamW  # noqa: E402
from bioscanclip.model.loss_func import ContrastiveLoss  # noqa: E402

from bioscanclip.hip import ops  # noqa: E402

GUARDS = {}
if os.environ.get("GUARD", "0") == "1":   # the f32 LoRA-gradient workspaces between sentinel bands: does a neighbour write past its end?
    _orig = ops.lora_grad_f32
    SENT = 12345.678

    def guarded(dqkv, y, M, H, lora_a, lora_b, dA, dB):
        key = (M, H, str(dqkv.device), torch.cuda.current_stream().cuda_stream)
        if key not in ops._LG32_WS:
            n = ops._l.load().bsclip_lora_grad_f32_workspace_floats(M, H)
            G = 1 << 18
            big = torch.full((n + 2 * G,), SENT, device=dqkv.device)
            ops._LG32_WS[key] = big[G:G + n]
            GUARDS[key] = (big, G, n)
        return _orig(dqkv, y, M, H, lora_a, lora_b, dA, dB)
    ops.lora_grad_f32 = guarded
    engine.ops.lora_grad_f32 = guarded


PERLAYER = {}   # GUARD=2: one workspace per (tower, layer) call, snapshotted after every step: which of t / dt / slabs goes wrong?
CALLS = {}
if os.environ.get("GUARD", "0") == "2":
    _orig2 = ops.lora_grad_f32

    def perlayer(dqkv, y, M, H, lora_a, lora_b, dA, dB):
        key = (M, H, str(dqkv.device), torch.cuda.current_stream().cuda_stream)
        L = 2 if H == 512 else 3
        i = CALLS.get(key, 0)
        CALLS[key] = (i + 1) % L
        if (key, i) not in PERLAYER:
            n = ops._l.load().bsclip_lora_grad_f32_workspace_floats(M, H)
            PERLAYER[(key, i)] = torch.zeros(n, device=dqkv.device)
        ops._LG32_WS[key] = PERLAYER[(key, i)]
        return _orig2(dqkv, y, M, H, lora_a, lora_b, dA, dB)
    ops.lora_grad_f32 = perlayer
    engine.ops.lora_grad_f32 = perlayer


def snap_perlayer():
    return {k: v.clone() for k, v in PERLAYER.items()}


def check_guards(tag):
    for key, (big, G, n) in GUARDS.items():
        lo, hi = (big[:G] != SENT).sum().item(), (big[G + n:] != SENT).sum().item()
        if lo or hi:
            print(f"  GUARD of workspace {key} damaged at {tag}: {lo} words below, {hi} above", flush=True)


steps = 8
TEXT = os.environ.get("TEXT", "1") == "1"
batches = [synth.synth_batch(16, seed=300 + s % 3, with_text=TEXT) for s in range(steps)]
cuda = lambda t: None if t is None else ({k: v.cuda() for k, v in t.items()} if isinstance(t, dict) else t.cuda())


def run(mode):
    model = _build(91, TEXT, False)
    opt = FusedAdamW(model.parameters(), lr=1e-3)
    opt.enable_device_hyper(True)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=3e-3, total_steps=steps, pct_start=0.3, anneal_strategy="cos", cycle_momentum=False)
    crit = ContrastiveLoss(torch.nn.CrossEntropyLoss(), 1 / 0.07)
    g = GraphedStep(model, opt, crit, warmup=2) if mode == "graph" else None
    losses, snaps, gsnaps, wsnaps = [], [], [], []
    for s in range(steps):
        image, dna, text, label = (cuda(t) for t in batches[s])
        if g is not None:
            loss = g(image, dna, text, label)
        else:
            opt.zero_grad()
            loss = crit(*model(image, dna, text), label)
            loss.backward()
            if opt.needs_attach():
                opt.attach(model)
            opt.step()
        sched.step()
        losses.append(loss.item())
        torch.cuda.synchronize()
        check_guards(f"{mode} step {s}")
        snaps.append({k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad})
        gsnaps.append({k: p.grad.detach().clone() for k, p in model.named_parameters() if p.requires_grad and p.grad is not None})
        wsnaps.append(snap_perlayer())
    WS_OF[mode] = wsnaps
    return losses, snaps, gsnaps


WS_OF = {}
ref = None
for it in range(int(os.environ.get("ITERS", "8"))):
    le, se, ge = run("eager")
    lg, sg, gg = run("graph")
    if ref is None:
        ref = (le, se, ge)
    ok = le == lg and all(torch.equal(se[-1][k], sg[-1][k]) for k in se[-1])
    same = lambda a, b: all(torch.equal(a[s_][k], b[s_][k]) for s_ in range(steps) for k in a[s_])
    print(f"iteration {it}: {'equal' if ok else 'MISMATCH'}; against iteration 0's eager run: eager {'same' if le == ref[0] and same(ge, ref[2]) else 'DEVIATES'}, "
          f"graph {'same' if lg == ref[0] and same(gg, ref[2]) else 'DEVIATES'}", flush=True)
    if ok:
        continue
    for s in range(steps):
        for (key, i), we in WS_OF["eager"][s].items():
            wg = WS_OF["graph"][s].get((key, i))
            if wg is None or torch.equal(we, wg):
                continue
            M, H = key[0], key[1]
            Mp = (M + 3) // 4 * 4
            segs = {"t": (0, 8 * Mp), "dt": (8 * Mp, 16 * Mp), "pa": (16 * Mp, 16 * Mp + 768 * 8 * H), "pb": (16 * Mp + 768 * 8 * H, we.numel())}
            for name, (a, b) in segs.items():
                d = (we[a:b] - wg[a:b])
                nbad = int((d != 0).sum().item())
                if nbad:
                    idx = torch.nonzero(d != 0).flatten()
                    print(f"    step {s}: workspace M={M} H={H} call {i} (layer {(2 if H == 512 else 3) - 1 - i}): segment {name} differs in {nbad} words, first at offset "
                          f"{int(idx[0])} last {int(idx[-1])} of {b - a}, max abs diff {d.abs().max().item():.3e}")
                    if nbad <= 400 and name in ("pa", "pb"):
                        slab = 8 * H
                        for o in idx[:12].tolist() + idx[-4:].tolist():
                            prev = [float(WS_OF["graph"][s2][(key, i)][a + o]) for s2 in range(max(0, s - 3), s)]
                            print(f"        word {o}: block {o // slab} element {o % slab} (k {(o % slab) // 64} lane {o % 64}): eager {float(we[a + o])!r} graph {float(wg[a + o])!r}"
                                  f"   graph at the steps before: {prev}")
                        blocks = sorted(set((idx // slab).tolist()))
                        print(f"        blocks touched: {blocks}")
                        ks = sorted(set(((idx % slab) // 64).tolist()))
                        print(f"        k indices touched: {ks}")
        bad_p = [k for k in se[s] if not torch.equal(se[s][k], sg[s][k])]
        bad_g = [k for k in ge[s] if k in gg[s] and not torch.equal(ge[s][k], gg[s][k])]
        print(f"  step {s}: loss eager {le[s]!r} graph {lg[s]!r} {'==' if le[s] == lg[s] else '!='}; params differing after the step: {len(bad_p)}"
              f" of {len(se[s])}; gradients differing: {len(bad_g)} of {len(ge[s])}")
        for k in bad_g[:6]:
            d = (ge[s][k] - gg[s][k]).abs().max().item()
            print(f"      grad {k}: max abs diff {d:.3e} (max abs {ge[s][k].abs().max().item():.3e})")
        for k in bad_p[:6]:
            d = (se[s][k] - sg[s][k]).abs().max().item()
            print(f"      param {k}: max abs diff {d:.3e}")
    if os.environ.get("STOP", "1") == "1":
        break
