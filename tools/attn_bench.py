"""Attention cores at the c2 size (BAIR B=64: 64 clips x 64 pixels x 8 heads), forward and backward, against their
ALGORITHMIC bytes (q, k, v [, dO] read once, o [dq, dk, dv] written once, fp32) - SURVEY 8(d).
Usage: python tools/attn_bench.py [--clips 64]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npvp_amd import ops
from npvp_amd.ops import AttnCfg

dev = "cuda:0"
N = int(sys.argv[sys.argv.index("--clips") + 1]) if "--clips" in sys.argv else 64
P, C, PEAK = 64, 512, 8000.0
torch.manual_seed(0)


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def report(name, nbytes, t):
    print(f"{name:52s} {t*1e6:8.1f} us  {nbytes/1e6:8.1f} MB algorithmic  {nbytes/t/1e9:7.0f} GB/s  {100*nbytes/t/1e9/PEAK:5.1f} % of 8 TB/s", flush=True)


for Tq, Tk, mask, pdrop in [(28, 28, 0, 0.1), (28, 2, 0, 0.1), (18, 18, 0, 0.1), (10, 10, 1, 0.1), (2, 2, 1, 0.1)]:
    Rq, Rk = N * Tq * P, N * Tk * P
    cfg = AttnCfg(1, N, P, 8, 0, Tq, Tk, 8, mask, pdrop)
    if Tq == Tk:       # self attention: q | k packed in one [R, 2C] projection output, as in the model
        qk = torch.randn(Rq, 2 * C, device=dev, requires_grad=True); v = torch.randn(Rk, C, device=dev, requires_grad=True)
        ins, fwd = [qk, v], lambda: ops.attn_packed(qk.detach(), v.detach(), cfg)
        y = ops.attn_packed(qk, v, cfg)
    else:
        q = torch.randn(Rq, C, device=dev, requires_grad=True); k = torch.randn(Rk, C, device=dev, requires_grad=True)
        v = torch.randn(Rk, C, device=dev, requires_grad=True)
        ins, fwd = [q, k, v], lambda: ops.attn(q.detach(), k.detach(), v.detach(), cfg)
        y = ops.attn(q, k, v, cfg)
    go = torch.randn_like(y)
    bq, bk = Rq * C * 4, Rk * C * 4
    report(f"temporal fwd  Tq={Tq} Tk={Tk} mask={mask}", 2 * bq + 2 * bk, timeit(fwd))
    report(f"temporal bwd  Tq={Tq} Tk={Tk} mask={mask}", 3 * bq + 4 * bk, timeit(lambda: torch.autograd.grad(y, ins, go, retain_graph=True)))

F_ = N * 30
R = F_ * P
qk = torch.randn(R, 2 * C, device=dev, requires_grad=True); v = torch.randn(R, C, device=dev, requires_grad=True)
cfg = AttnCfg(0, F_, 64, 8, 4, 0, 0, 8, 0, 0.1)
y = ops.attn_packed(qk, v, cfg); go = torch.randn_like(y)
report("spatial 4x4 windows fwd (1920 frames)", 4 * R * C * 4, timeit(lambda: ops.attn_packed(qk.detach(), v.detach(), cfg)))
report("spatial 4x4 windows bwd", 7 * R * C * 4, timeit(lambda: torch.autograd.grad(y, [qk, v], go, retain_graph=True)))
