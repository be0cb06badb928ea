"""Frame statistics in the GEMM epilogue vs a separate statistics pass (c2 decoder size): timing only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npvp_amd import ops
from npvp_amd._lib import lib, check
dev = "cuda:0"
R = 114688
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for N, K in [(2048, 512), (512, 2048)]:
    x = torch.randn(R, K, device=dev); w = torch.nn.Parameter(torch.randn(N, K, device=dev) / K ** 0.5); b = torch.randn(N, device=dev)
    part = torch.empty((R // 64) * (N // 64) * 2, device=dev)
    y = ops.linear_fwd(x, w, b)
    mean = torch.empty(R // 64, device=dev); rstd = torch.empty_like(mean)
    t_plain = timeit(lambda: ops.linear_fwd(x, w, b))
    t_rs = timeit(lambda: ops.linear_fwd(x, w, b, rowstats=part))
    t_fs = timeit(lambda: check(lib().npvp_frame_stats(ops._ptr(y), ops._p(0), ops._ptr(mean), ops._ptr(rstd), R // 64, 1, 64 * N, 1e-5, ops._stream()), "fs"))
    print(f"[{R} x {K}] -> {N}: plain {t_plain:.0f} us, with epilogue statistics {t_rs:.0f} us (+{t_rs - t_plain:.0f}), separate frame_stats pass {t_fs:.0f} us")
