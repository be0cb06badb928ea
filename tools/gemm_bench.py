"""GEMM micro-benchmark on the predictor's shapes (R token rows x the linears' (N_out, K_in)): algorithmic TFLOP/s of the
forward / dgrad / weight-gradient GEMMs and max error against an fp64 product.
Usage: python tools/gemm_bench.py [--rows R] [--iters N] [--mode bf16x6|f32] [--check]"""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npvp_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, nargs="*", default=[114688, 20480, 8192])      # c2 decoder, c1, c2 encoder / c4 shard
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--mode", default="bf16x6")
ap.add_argument("--check", action="store_true")
ap.add_argument("--grad-scale", type=float, default=1.0, help="magnitude of dy (gradient-sized operands: 1e-8)")
ap.add_argument("--heavy", action="store_true", help="log-normal magnitudes on dy (heavy tail) and per-row scales over 1e-3..1e3")
args = ap.parse_args()
dev = "cuda:0"
shapes = [(512, 512), (1024, 512), (2048, 512), (512, 2048), (512, 1024)]      # (N_out, K_in) of the linears
torch.manual_seed(0)
ops.set_gemm_precision(args.mode)


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters * 1e-3


def err(a, ref):
    return float((a.double() - ref).norm() / ref.norm())


for R in args.rows:
    tot_t, tot_f = [0.0, 0.0, 0.0], 0.0
    for N, K in shapes:
        x = torch.randn(R, K, device=dev); w = torch.nn.Parameter(torch.randn(N, K, device=dev) / K ** 0.5)
        dy = torch.randn(R, N, device=dev) * args.grad_scale; b = torch.randn(N, device=dev)
        if args.heavy:
            dy = dy * torch.exp(2 * torch.randn(R, N, device=dev)) * torch.exp(torch.empty(R, 1, device=dev).uniform_(-7, 7))
        fl = 2.0 * R * N * K
        t1 = timeit(lambda: ops.linear_fwd(x, w, b))
        t2 = timeit(lambda: ops.linear_dgrad(dy, w))
        t3 = timeit(lambda: ops.linear_wgrad(dy, x))
        for i, t in enumerate((t1, t2, t3)):
            tot_t[i] += t
        tot_f += fl
        extra = ""
        if args.check:
            n = min(R, 4096)
            e1 = err(ops.linear_fwd(x, w, b)[:n], x[:n].double() @ w.double().T + b.double())
            e2 = err(ops.linear_dgrad(dy, w)[:n], dy[:n].double() @ w.double())
            dgr, ref = ops.linear_dgrad(dy, w)[:n].double(), dy[:n].double() @ w.double()
            e2r = float(((dgr - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-300)).max())
            m = min(R, 16384)
            e3 = err(ops.linear_wgrad(dy[:m].contiguous(), x[:m].contiguous()), dy[:m].double().T @ x[:m].double())
            t32 = err((x[:n] @ w.T + b), x[:n].double() @ w.double().T + b.double())
            extra = f"  err fwd {e1:.1e} (torch fp32 {t32:.1e}) dgrad {e2:.1e} (worst row {e2r:.1e}) wgrad[{m}] {e3:.1e}"
        print(f"R={R:6d} N={N:5d} K={K:5d}  fwd {fl/t1/1e12:6.1f} TF ({t1*1e6:7.1f} us)  dgrad {fl/t2/1e12:6.1f} TF ({t2*1e6:7.1f} us)  "
              f"wgrad {fl/t3/1e12:6.1f} TF ({t3*1e6:7.1f} us){extra}", flush=True)
    print(f"R={R:6d} all shapes: fwd {tot_f/tot_t[0]/1e12:.1f}  dgrad {tot_f/tot_t[1]/1e12:.1f}  wgrad {tot_f/tot_t[2]/1e12:.1f} TF", flush=True)
