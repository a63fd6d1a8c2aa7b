"""Does intra-tower concurrency pay?  Two independent SimpleCLIP replicas at local batch B/2, each with its own main stream and
its own pair of tower streams (four tower streams in flight), enqueued back to back by one Python thread, against one replica at
batch B (bench.py's configuration, eager).  Same total samples per iteration; the loss differs (two B/2 x B/2 matrices instead of
one B x B), which is irrelevant to the timing of the towers that this probe is about.

    python tools/microbatch_probe.py [B] [iters]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))

import bench  # noqa: E402


def make(B, device, seed):
    from bioscanclip.hip.optim import FusedAdamW
    from bioscanclip.model.loss_func import ContrastiveLoss
    model = bench.build_model(False, device, seed=seed).train()
    image, dna, _ = bench.synthetic_batch(B, False, device, seed=seed)
    label = torch.arange(B, device=device)
    opt = FusedAdamW(model.parameters(), lr=1e-3)
    crit = ContrastiveLoss(torch.nn.CrossEntropyLoss(), 1 / 0.07)

    def step():
        opt.zero_grad()
        io, do, to = model(image, dna, None)
        loss = crit(io, do, to, label)
        loss.backward()
        if opt.needs_attach():
            opt.attach(model)
        opt.step()
        return loss
    return step


def timeit(fn, iters):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / iters * 1e3, (t1 - t0) / iters * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    device = torch.device("cuda", 0)
    from bioscanclip.model import simple_clip as sc

    one = make(B, device, 1)
    ms, enq = timeit(one, iters)
    print(f"one replica, B={B}: {ms:.2f} ms/iter (host enqueue {enq:.2f})", flush=True)
    del one
    torch.cuda.empty_cache()

    halves = []
    for i in range(2):
        halves.append((make(B // 2, device, 10 + i), torch.cuda.Stream(device=device), {}))

    def both():
        cur = torch.cuda.current_stream()
        for step, main_stream, streams in halves:
            sc._streams = streams                    # this replica's own tower streams
            main_stream.wait_stream(cur)
            with torch.cuda.stream(main_stream):
                step()
        for _, main_stream, _ in halves:
            cur.wait_stream(main_stream)

    ms, enq = timeit(both, iters)
    print(f"two replicas, B={B // 2} each, concurrent: {ms:.2f} ms/iter (host enqueue {enq:.2f})", flush=True)

    def serial():
        for step, _, streams in halves:
            sc._streams = streams
            step()

    ms, enq = timeit(serial, iters)
    print(f"two replicas, B={B // 2} each, one after the other: {ms:.2f} ms/iter (host enqueue {enq:.2f})", flush=True)


if __name__ == "__main__":
    main()
