"""Forward-only throughput of the HIP encoders (the feature-extraction half of the retrieval evaluation):
eval mode, no autograd, L2-normalised features left in HBM.  python tools/inference_bench.py [--batch 256] [--text]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bioscan-clip_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--text", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = bench.build_model(a.text, dev).eval()
    image, dna, text = bench.synthetic_batch(a.batch, a.text, dev, seed=1)
    with torch.no_grad():
        for _ in range(3):
            out = model(image, dna, text)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = model(image, dna, text)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"metric": "forward-only pairs/s (eval mode, I+D%s)" % ("+T" if a.text else ""),
                      "value": round(a.batch / dt, 1), "ms_per_batch": round(dt * 1e3, 3), "batch": a.batch,
                      "unit_norm_check": float(out[0].norm(dim=-1).mean())}))


if __name__ == "__main__":
    main()
