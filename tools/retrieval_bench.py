"""Measures the retrieval row (bsclip_topk_ip) at BIOSCAN-1M evaluation size: 21 118 keys, 768-d, top-5.

    python tools/retrieval_bench.py [--queries 16384] [--cpu-queries 512]

Prints one JSON line: queries/s on the GPU (features resident in HBM), the select kernel's share, and the numpy oracle's
rate on a bounded sample (checker timed beside the product, never part of it).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))

from bioscanclip.hip import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=16384)
    ap.add_argument("--keys", type=int, default=21118)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--cpu-queries", type=int, default=512)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    keys = torch.randn(a.keys, a.dim, device="cuda", generator=g)
    q = torch.randn(a.queries, a.dim, device="cuda", generator=g)
    for _ in range(2):
        ops.topk_ip(q, keys, a.k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        sims, idx = ops.topk_ip(q, keys, a.k)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    # algorithmic work: 2*Q*K*4D flops on MFMA (4-term split), Q*Kp*4 B written + read for the score slab
    kp = (a.keys + 255) // 256 * 256
    out = {"metric": "retrieval_queries_per_sec", "value": a.queries / ms * 1e3, "ms": ms, "queries": a.queries,
           "keys": a.keys, "dim": a.dim, "k": a.k,
           "gemm_tflops_at_total_time": 2 * a.queries * kp * 4 * a.dim / ms / 1e9,
           "score_slab_GBps_at_total_time": 2 * a.queries * kp * 4 / ms / 1e6}
    if a.cpu_queries:
        from oracle import retrieval as R
        qc, kc = q[: a.cpu_queries].cpu().numpy(), keys.cpu().numpy()
        t = time.perf_counter()
        ref_s, ref_i = R.topk_ip(qc, kc, a.k)
        dt = time.perf_counter() - t
        out["cpu_oracle_queries_per_sec"] = a.cpu_queries / dt
        out["cpu_sample"] = f"{a.cpu_queries} queries x {a.keys} keys, numpy f64"
        out["indices_equal_on_sample"] = bool((idx[: a.cpu_queries].cpu().numpy() == ref_i).all())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
