"""GEMM micro-benchmark on the predictor's c1 shapes (R = 20480 token rows): TFLOP/s per variant and mode.
Usage: python tools/gemm_bench.py [modes...] [--iters N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npvp_amd import ops

dev = "cuda:0"
modes = [a for a in sys.argv[1:] if not a.startswith("--")] or ["f32", "bf16x3", "bf16x6"]
dbg = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--dbg=")]
dbg = dbg[0] if dbg else 0
iters = 20
R = 20480
shapes = [(512, 512), (1024, 512), (2048, 512), (512, 2048), (512, 1024)]      # (N_out, K_in) of the linears
torch.manual_seed(0)


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for mode in modes:
    ops.set_gemm_precision(mode)
    ops.GEMM_PRECISION |= dbg << 8        # ablation flags: 1 no global loads, 2 no split/stage, 4 no MFMA (results invalid)
    tot_t, tot_f = 0.0, 0.0
    for N, K in shapes:
        x = torch.randn(R, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
        dy = torch.randn(R, N, device=dev); b = torch.randn(N, device=dev)
        fl = 2.0 * R * N * K
        t1 = timeit(lambda: ops.linear_fwd(x, w, b))
        t2 = timeit(lambda: ops.linear_dgrad(dy, w))
        t3 = timeit(lambda: ops.linear_wgrad(dy, x))
        tot_t += t1 + t2 + t3; tot_f += 3 * fl
        print(f"{mode:7s} N={N:5d} K={K:5d}  fwd {fl/t1/1e12:7.1f} TF ({t1*1e6:7.1f} us)  dgrad {fl/t2/1e12:7.1f} TF ({t2*1e6:7.1f} us)  "
              f"wgrad {fl/t3/1e12:7.1f} TF ({t3*1e6:7.1f} us)", flush=True)
    print(f"{mode:7s} all shapes: {tot_f/tot_t/1e12:.1f} TF")
