"""Fixed-vs-per-K cost of the forward GEMM: time over K at fixed [R, N] (tools/gemm_bench.py has the real shapes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npvp_amd import ops
dev = "cuda:0"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 114688
Ns = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [512, 2048]
def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for N in Ns:
    for K in (32, 64, 128, 256, 512, 1024, 2048):
        x = torch.randn(R, K, device=dev); w = torch.nn.Parameter(torch.randn(N, K, device=dev)); b = torch.randn(N, device=dev)
        t = timeit(lambda: ops.linear_fwd(x, w, b))
        t0 = timeit(lambda: ops.linear_fwd(x, w, None))
        print(f"R={R} N={N} K={K:5d}: {t:8.1f} us (no bias {t0:8.1f})  {2.0*R*N*K/t/1e6:6.1f} TF", flush=True)
