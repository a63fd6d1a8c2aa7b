"""Shared test helpers (fingerprints matching oracle/gen_golden.py:summary)."""
import json
import os

import torch

from oracle import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLD, name + ".json")) as f:
        return json.load(f)


def summary(key, t):
    t = t.detach().to("cpu", torch.float64).reshape(-1)
    probe = synth.synth_tensor("probe." + key, t.shape, seed=7).to(torch.float64) / 0.02
    return {"shape": list(t.shape), "norm": t.norm().item(), "sum": t.sum().item(),
            "first": t[:8].tolist(), "probe": (t * probe).sum().item()}


def check_summary(key, t, gold, rtol, what="", first_slack=1.0, probe=True):
    """Compare a tensor against a golden fingerprint.  ``norm`` is compared relatively; ``first`` and ``probe``
    (a random projection, i.e. a checksum sensitive to every element) relative to the tensor's norm."""
    s = summary(key, t)
    assert s["shape"] == gold["shape"], f"{what}{key}: shape {s['shape']} vs {gold['shape']}"
    n = max(gold["norm"], 1e-30)
    assert abs(s["norm"] - gold["norm"]) <= rtol * n, f"{what}{key}: norm {s['norm']} vs {gold['norm']}"
    numel = 1
    for d in gold["shape"]:
        numel *= d
    # probe ~ N(0,1) entries: |probe . err| ~ ||err||; elementwise first-8 error ~ ||err|| / sqrt(numel)
    if probe:
        assert abs(s["probe"] - gold["probe"]) <= 4 * rtol * n, f"{what}{key}: probe {s['probe']} vs {gold['probe']}"
    tol_first = first_slack * 6 * rtol * n / numel ** 0.5 + 1e-12
    for a, b in zip(s["first"], gold["first"]):
        assert abs(a - b) <= tol_first, f"{what}{key}: first {s['first']} vs {gold['first']} (tol {tol_first})"
    return s


def summary_distance(key, t, gold):
    """Measured distances to a golden fingerprint, in the units ``check_summary`` asserts on (all relative to the golden
    norm): norm difference, projection difference / 4, worst of the first-8 differences / (6 / sqrt(numel))."""
    s = summary(key, t)
    n = max(gold["norm"], 1e-30)
    numel = 1
    for d in gold["shape"]:
        numel *= d
    return {"norm": abs(s["norm"] - gold["norm"]) / n, "probe": abs(s["probe"] - gold["probe"]) / (4 * n),
            "first": max(abs(a - b) for a, b in zip(s["first"], gold["first"])) * numel ** 0.5 / (6 * n)}


def rel_err(a, b):
    a = a.detach().to("cpu", torch.float64)
    b = b.detach().to("cpu", torch.float64)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


import contextlib  # noqa: E402


@contextlib.contextmanager
def skip_param_init():
    """Build a model without running its default parameter initialisers (2-3 s of ``normal_`` / ``kaiming_uniform_`` per
    full-depth tower on the test box's CPU share).  ONLY for models whose whole ``state_dict`` is loaded right afterwards
    (``oracle.synth.synth_state_dict`` + strict ``load_state_dict``): until then the parameters are uninitialised memory."""
    import torch.nn.init as init
    names = ["normal_", "trunc_normal_", "uniform_", "kaiming_uniform_", "kaiming_normal_", "xavier_uniform_", "xavier_normal_",
             "zeros_", "ones_", "constant_"]
    saved = {n: getattr(init, n) for n in names}
    for n in names:
        setattr(init, n, lambda t, *a, **k: t)
    try:
        yield
    finally:
        for n, f in saved.items():
            setattr(init, n, f)
