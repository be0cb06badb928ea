"""The six-term bf16 arithmetic (round 2's default, `--gemm bf16x6`) still agrees with fp64 on the path's GEMM shapes and runs a
training step: python tools/bf16x6_check.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npvp_amd import ops
ops.set_gemm_precision("bf16x6")
dev = "cuda:0"
torch.manual_seed(0)
worst = 0.0
for R in (8192, 320):
    for N, K in ((512, 512), (2048, 512), (512, 2048)):
        x = torch.randn(R, K, device=dev); w = torch.nn.Parameter(torch.randn(N, K, device=dev) / K ** 0.5)
        dy = torch.randn(R, N, device=dev); b = torch.randn(N, device=dev)
        e1 = float((ops.linear_fwd(x, w, b).double() - (x.double() @ w.double().T + b.double())).norm() / (x.double() @ w.double().T).norm())
        e2 = float((ops.linear_dgrad(dy, w).double() - dy.double() @ w.double()).norm() / (dy.double() @ w.double()).norm())
        g = ops.linear_wgrad(dy, x, True)
        e3 = float((g[0].double() - dy.double().T @ x.double()).norm() / (dy.double().T @ x.double()).norm())
        e4 = float((g[1].double() - dy.double().sum(0)).norm() / dy.double().sum(0).norm())
        worst = max(worst, e1, e2, e3, e4)
        print(f"R={R} N={N} K={K}: fwd {e1:.2e} dgrad {e2:.2e} wgrad {e3:.2e} bias-grad {e4:.2e}")
assert worst < 2e-6, worst
print("bf16x6 GEMMs ok, worst", worst)
