import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "bioscan-clip_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _reset_process_global_kernel_switches(request):
    """The library's benchmarking switches (GEMM tile override, persistent grid, the second implementations of the InfoNCE and the
    exact-mode attention kernels) are process-global: a GPU test that fails between setting one and resetting it must not change which
    kernel every later test exercises.  Reset after every GPU test, whatever its outcome."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import torch
    if not torch.cuda.is_available():
        return
    from bioscanclip.hip import ops
    ops.set_gemm_tile(0)
    ops.set_gemm_persistent_grid(0)
    ops.infonce_set_impl(0)
    ops.exact_attn_set_impl(0)
