"""Accuracy / range probe of the fp16 two-term split GEMM (mode f16x3) against float64, next to bf16x6db.
Usage: python tools/h3_check.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npvp_amd import ops

dev = "cuda:0"
torch.manual_seed(0)
R, N, K = 4096 + 64, 512 + 128, 2048       # ragged tile edges on purpose


def rel(a, ref):
    return float((a.double() - ref).norm() / ref.norm())


for scale_x in (1.0, 1e-2, 1e-4, 1e-5, 1e-6, 1e-8, 1e3):
    x = torch.randn(R, K, device=dev) * scale_x
    # wide dynamic range inside the tensor too
    x = x * torch.exp(torch.randn(R, 1, device=dev) * 2.0)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    dy = torch.randn(R, N, device=dev) * scale_x
    ref_f = x.double() @ w.double().t()
    ref_d = dy.double() @ w.double()
    ref_w = dy.double().t() @ x.double()
    line = f"scale {scale_x:8.0e}: "
    for mode in ([a for a in sys.argv[1:] if not a.startswith("-")] or ["bf16x6db", "f16x3"]):
        ops.set_gemm_precision(mode)
        yf = ops.linear_fwd(x, w, None); yd = ops.linear_dgrad(dy, w); yw = ops.linear_wgrad(dy, x)
        yw = yw[0] if isinstance(yw, tuple) else yw
        line += f"{mode}: fwd {rel(yf, ref_f):.2e} dgrad {rel(yd, ref_d):.2e} wgrad {rel(yw, ref_w):.2e} | "
    print(line, flush=True)
