"""What a dropout / DropPath mask in the epilogue costs the f16x3 GEMM: [R x K] x [N x K]^T at the c2 decoder size, residual
epilogue, mask off / per element (mode 0) / per row group (mode 1).  Usage: python tools/gemm_drop_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npvp_amd import ops
from npvp_amd.ops import Drop, NO_DROP

dev = "cuda:0"
torch.manual_seed(0)


def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for R, K, N in [(114688, 512, 512), (114688, 2048, 512), (8192, 512, 512)]:
    x = torch.randn(R, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.zeros(N, device=dev)
    res = torch.randn(R, N, device=dev)
    for name, d in [("no mask", NO_DROP), ("element mask", Drop(0.1, 0)), ("row-group mask", Drop(0.1, 1, 28 * 64, R // (28 * 64) or 1))]:
        t = timeit(lambda: ops.linear_fwd(x, w, b, residual=res, drop=d))
        print(f"fwd  R={R} K={K} N={N} residual + {name:15s} {t:8.1f} us  {2.0*R*K*N/t/1e6:6.1f} TF", flush=True)
    dy = torch.randn(R, N, device=dev)
    aux = torch.randn(R, K, device=dev)
    for name, d in [("no mask", NO_DROP), ("element mask", Drop(0.1, 0))]:
        t = timeit(lambda: ops.linear_dgrad(dy, w, act=3, aux_in=aux, drop=d))
        print(f"dgrad R={R} K={N} N={K} GELU' + {name:15s} {t:8.1f} us  {2.0*R*K*N/t/1e6:6.1f} TF", flush=True)
    y = torch.randn(R, N, device=dev)
    for name, d in [("element mask", Drop(0.1, 0)), ("row-group mask", Drop(0.1, 1, 28 * 64, R // (28 * 64) or 1))]:
        t = timeit(lambda: ops.drop_apply(y, d))
        print(f"drop_apply [{R} x {N}] {name:15s} {t:8.1f} us  {2*R*N*4/t/1e6:6.2f} TB/s", flush=True)
