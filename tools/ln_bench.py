"""LayerNorm forward / backward at the step's ViT shape (M = 50 432, H = 768), plain and with the LoRA-A projection / dt . A term."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bioscan-clip_amd"))
import torch  # noqa: E402

from bioscanclip.hip import ops  # noqa: E402

M, H = int(os.environ.get("M", 50432)), 768
x = torch.randn(M, H, device="cuda")
g, b = torch.randn(H, device="cuda"), torch.randn(H, device="cuda")
A = torch.randn(8, H, device="cuda") * 0.1
y = torch.empty(M, H + 64, device="cuda", dtype=torch.bfloat16)
st = torch.empty(M, 2, device="cuda")
gg = torch.randn(M, H, device="cuda").bfloat16()
gr = torch.randn(M, H, device="cuda")
dt = torch.randn(M, 8, device="cuda")
dx, dxb = torch.empty(M, H, device="cuda"), torch.empty(M, H, device="cuda", dtype=torch.bfloat16)


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"fwd plain {t(lambda: ops.layernorm_fwd(x, g, b, 1e-6, y_bf16=y, stats=st)):.1f} us   "
      f"fwd + LoRA t {t(lambda: ops.layernorm_fwd(x, g, b, 1e-6, y_bf16=y, lora_a=A, stats=st)):.1f} us")
print(f"bwd plain {t(lambda: ops.layernorm_bwd(x, st, g, 0, g_resid=gr, g_gemm=gg, dx_f32=dx, dx_bf16=dxb)):.1f} us   "
      f"bwd + dt.A {t(lambda: ops.layernorm_bwd(x, st, g, 0, g_resid=gr, g_gemm=gg, dt=dt, lora_a=A, dx_f32=dx, dx_bf16=dxb)):.1f} us")
# the step's dtypes: bf16 residual stream in, bf16 residual-gradient stream, one bf16 output (profiles/r05_f_serial_kernel_stats.csv:
# layernorm_fwd_kernel<768, true, true> 35.4 us, <768, true, false> 23.8; layernorm_bwd_kernel<768, true, true> 59.2, <.., false> 45.1)
xb, grb = x.bfloat16(), gr.bfloat16()
print(f"bf16 stream: fwd plain {t(lambda: ops.layernorm_fwd(xb, g, b, 1e-6, y_bf16=y, stats=st)):.1f} us   "
      f"fwd + LoRA t {t(lambda: ops.layernorm_fwd(xb, g, b, 1e-6, y_bf16=y, lora_a=A, stats=st)):.1f} us")
print(f"bf16 stream: bwd plain {t(lambda: ops.layernorm_bwd(xb, st, g, 0, g_resid=grb, g_gemm=gg, dx_bf16=dxb)):.1f} us   "
      f"bwd + dt.A {t(lambda: ops.layernorm_bwd(xb, st, g, 0, g_resid=grb, g_gemm=gg, dt=dt, lora_a=A, dx_bf16=dxb)):.1f} us")
