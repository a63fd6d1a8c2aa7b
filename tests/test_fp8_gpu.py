"""fp8 GEMM family (BASELINE.json configs[4]) against exact emulation: the products of two e4m3 values are exact in f32, so the
kernel must agree with a float matmul of the DEQUANTISED operands to accumulation-order precision -- for both MFMA forms."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import rel_err  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bioscanclip.hip import ops as o
    o.init_tables()
    return o


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def gelu(x):
    return torch.nn.functional.gelu(x)


@pytest.mark.parametrize("form", [1, 2])
@pytest.mark.parametrize("M,N,K,aug", [(300, 256, 128, False), (1000, 768, 768, True), (2600, 2304, 768, True),
                                       (5120, 768, 3072, False), (197 * 16, 3072, 768, False)])
def test_gemm_fp8_matches_dequantised_matmul(ops, form, M, N, K, aug):
    from bioscanclip.hip.lib import EPI_BF16, EPI_F32, EPI_GELU_FP8, EPI_RESID_F32
    FP8 = ops.FP8
    a = rnd(M, K, seed=1).cuda()
    a8 = a.to(FP8)                                            # activations: scale 1 (LayerNorm / GELU outputs are O(1))
    w = rnd(N, K, seed=2, scale=0.05).cuda()
    w8, ws = ops.quantize_rows_fp8(w.contiguous())
    # the quantiser itself: row amax -> 448, round to nearest e4m3
    ref_s = w.abs().amax(dim=1) / 448
    assert torch.allclose(ws, ref_s, rtol=1e-6)
    assert torch.equal(w8.view(torch.uint8), (w / ref_s[:, None]).to(FP8).view(torch.uint8))
    bias = rnd(N, seed=3).cuda()
    alpha = ws.clone()                                        # activation scale 1 x weight-row scale
    ref = (a8.float() @ w8.float().t()) * alpha + bias
    kw = {}
    if aug:
        t = torch.zeros(M, 64, dtype=torch.bfloat16, device="cuda")
        t[:, :8] = rnd(M, 8, seed=4, scale=0.3).cuda().bfloat16()
        b = torch.zeros(N, 64, dtype=torch.bfloat16, device="cuda")
        b[:, :8] = (rnd(N, 8, seed=5, scale=0.05).cuda() / alpha[:, None]).bfloat16()
        ref = ref + (t.float() @ b.float().t()) * alpha
        kw = dict(a_aug=t, b_aug=b)
    out32 = torch.full((M + 3, N), float("nan"), device="cuda")
    ops.gemm_fp8(a8, w8, out32, alpha, bias, EPI_F32, M=M, form=form, **kw)
    assert rel_err(out32[:M], ref) < 2e-5
    assert torch.isnan(out32[M:]).all()
    out16 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_fp8(a8, w8, out16, alpha, bias, EPI_BF16, form=form, **kw)
    assert rel_err(out16.float(), ref) < 4e-3
    r = rnd(M, N, seed=6).cuda()
    ops.gemm_fp8(a8, w8, out32, alpha, bias, EPI_RESID_F32, resid=r, M=M, form=form, **kw)
    assert rel_err(out32[:M], ref + r) < 2e-5
    out8 = torch.empty(M, N, dtype=FP8, device="cuda")
    z = torch.empty(M, N, dtype=torch.uint8, device="cuda")
    ops.gemm_fp8(a8, w8, out8, alpha, bias, EPI_GELU_FP8, aux=z, form=form, **kw)
    want = gelu(ref)
    # e4m3 has 3 mantissa bits: compare codes -- equal, or one code apart where the value sits on a rounding boundary
    got, exp = out8.float(), want.to(FP8).float()
    assert ((got - exp).abs() <= 0.126 * exp.abs() + 2e-3).all()
    assert (got != exp).float().mean().item() < 2e-3
    x = ref.double()
    dg = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
    assert ((z.float() * (1.26 / 255) - 0.13) - dg.float()).abs().max().item() < 0.5 * 1.26 / 255 + 3e-4


def test_gemm_fp8_forms_agree_and_integer_exact(ops):
    """Small integers are exact in e4m3 and their dot products exact in f32: both MFMA forms must return the integer matrix
    product exactly (catches any k-slot mismatch between the A and B fragments)."""
    from bioscanclip.hip.lib import EPI_F32
    FP8 = ops.FP8
    g = torch.Generator().manual_seed(7)
    M, N, K = 512, 256, 384
    a = torch.randint(-4, 5, (M, K), generator=g).float().cuda()
    w = torch.randint(-3, 4, (N, K), generator=g).float().cuda()
    w[:, ::7] = 0          # asymmetric structure along k
    ref = a @ w.t()
    one, zero = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    for form in (1, 2):
        out = torch.empty(M, N, device="cuda")
        ops.gemm_fp8(a.to(FP8), w.to(FP8), out, one, zero, EPI_F32, form=form)
        assert torch.equal(out, ref), form
