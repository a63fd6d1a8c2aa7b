"""Where the collectives of the captured W > 1 step sit in time (hip/graph.py GraphedDistStep: per-tower graphs, each tower's
all-gather / all-reduce issued from that tower's stream).  One-GPU rehearsal through the REAL process group:

    BSCLIP_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29661 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python tools/dist_overlap_probe.py

An instrumented copy of ``_replay``: HIP timing events on the tower streams around every graph launch, and -- on a probe stream
per collective that does nothing but ``work.wait()`` -- an event that fires when that collective has completed.  Printed per
step: the window of every tower's forward / backward graph and the completion time of every collective, in ms from the step's
start.  A collective "overlaps" when it completes before another tower's graph window closes: it ran beside that tower's kernels
and the loss / optimizer graph that waits for it starts no later than it would without the collective."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bioscan-clip_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402

assert os.environ.get("BSCLIP_FORCE_DIST") == "1", "run with BSCLIP_FORCE_DIST=1 (see the docstring)"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from bioscanclip.hip.graph import GraphedDistStep  # noqa: E402
from bioscanclip.hip.optim import FusedAdamW  # noqa: E402
from bioscanclip.model.loss_func import GlobalBatchContrastiveLoss  # noqa: E402

B = int(os.environ.get("B", "256"))
dev = torch.device("cuda", 0)
model = bench.build_model(True, dev).train()
image, dna, text = bench.synthetic_batch(B, True, dev, seed=1234)
label = torch.arange(B, device=dev)
opt = FusedAdamW(model.parameters(), lr=1e-3)
g = GraphedDistStep(model, opt, GlobalBatchContrastiveLoss(torch.nn.CrossEntropyLoss(), 1 / 0.07), warmup=2)
for _ in range(6):
    g(image, dna, text, label)
torch.cuda.synchronize()
assert g.towers is not None


def ev():
    return torch.cuda.Event(enable_timing=True)


def probe(work, stream, events, key):
    with torch.cuda.stream(stream):
        work.wait()                      # this stream (and only it) waits for the collective
        e = ev()
        e.record()
        events[key] = e


_all = [torch.cuda.Stream() for _ in range(16)]
_sh = int(os.environ.get('PROBE_SHIFT', '0'))      # which of the pooled streams carry the probes (HIP maps streams onto few hardware queues)
probes = _all[_sh:_sh + 8]
for step in range(3):
    g.optimizer.advance_host_state()
    E = {}
    main = torch.cuda.current_stream()
    E["start"] = ev()
    E["start"].record(main)
    works = [dist.all_gather_into_tensor(g.labels_full, label.contiguous(), group=g.group, async_op=True)]
    for i, t in enumerate(g.towers):
        t.stream.wait_stream(main)
        with torch.cuda.stream(t.stream):
            a, b = ev(), ev()
            a.record()
            t.gF.replay()
            b.record()
            E[f"{t.name} forward"] = (a, b)
    order = sorted(g.towers, key=lambda t: g._ORDER[t.name])     # as GraphedDistStep._replay: shortest tower first
    for i, t in enumerate(order):
        with torch.cuda.stream(t.stream):
            w = dist.all_gather_into_tensor(t.full, t.emb.detach(), group=g.group, async_op=True)
            works.append(w)
        probe(w, probes[i], E, f"{t.name} all-gather done")
    for w in works:
        w.wait()
    for t in g.towers:
        main.wait_stream(t.stream)
    a, b = ev(), ev()
    a.record(main)
    g.gL.replay()
    b.record(main)
    E["loss"] = (a, b)
    for i, t in enumerate(g.towers):
        t.work = None
        if t.gB is None:
            continue
        t.stream.wait_stream(main)
        with torch.cuda.stream(t.stream):
            a, b = ev(), ev()
            a.record()
            t.gB.replay()
            b.record()
            E[f"{t.name} backward"] = (a, b)
    for i, t in enumerate(order):
        if t.gB is not None and t.flat is not None:
            with torch.cuda.stream(t.stream):
                t.work = dist.all_reduce(t.flat.grad, op=dist.ReduceOp.SUM, group=g.group, async_op=True)
            probe(t.work, probes[4 + i], E, f"{t.name} all-reduce done")
    for t in g.towers:
        if t.work is not None:
            t.work.wait()
        main.wait_stream(t.stream)
    a, b = ev(), ev()
    a.record(main)
    g.gC.replay()
    b.record(main)
    E["AdamW"] = (a, b)
    torch.cuda.synchronize()
    t0 = E.pop("start")
    print(f"step {step} (I+D+T, local batch {B}, world_size 1 through RCCL): ms from the step's start")
    for k, v in E.items():
        if isinstance(v, tuple):
            print(f"    {k:26s} {t0.elapsed_time(v[0]):8.3f} .. {t0.elapsed_time(v[1]):8.3f}")
        else:
            print(f"    {k:26s} {'':8s}    {t0.elapsed_time(v):8.3f}")
dist.destroy_process_group()
