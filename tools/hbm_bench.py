"""Achieved HBM bandwidth of the non-GEMM kernels at the c1 size (one encoder pass: 320 frames = 20480 token rows),
against their ALGORITHMIC bytes (each input read once, each output written once, fp32) - SURVEY 8(d).
Usage: python tools/hbm_bench.py [--frames 320]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npvp_amd import ops
from npvp_amd._lib import lib

dev = "cuda:0"
F_ = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 320
R, C, HID, P = F_ * 64, 512, 2048, 64
PEAK = 8000.0
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
    print(f"{name:44s} {t*1e6:8.1f} us  {nbytes/1e6:8.1f} MB algorithmic  {nbytes/t/1e9:7.0f} GB/s  {100*nbytes/t/1e9/PEAK:5.1f} % of 8 TB/s", flush=True)


h = torch.randn(F_, P * HID, device=dev)
g = torch.randn(F_, P * HID, device=dev)
w = torch.randn(P * HID, device=dev); b = torch.randn(P * HID, device=dev)
nb = h.numel() * 4
L = lib()
mean = torch.empty(F_, device=dev); rstd = torch.empty(F_, device=dev)
report("frame_stats [F, 64*2048]", nb, timeit(lambda: ops.check(L.npvp_frame_stats(ops._ptr(h), ops._p(0), ops._ptr(mean), ops._ptr(rstd), F_, 1, P * HID, 1e-5, ops._stream()), "s")))
hh = h.clone().requires_grad_()
report("frame-LN+GELU fwd (stats + apply)", 3 * nb, timeit(lambda: ops.frameln_act(hh.detach(), w, b, None, F_)))
out = ops.frameln_act(hh, w, b, None, F_)
report("frame-LN+GELU bwd (stats + dh,dw,db)", 5 * nb, timeit(lambda: torch.autograd.grad(out, hh, g, retain_graph=True)))
wtb = torch.randn(10, HID, device=dev)
a = h.view(F_, P, HID)
report("depthwise 3x3 fwd", 2 * nb, timeit(lambda: ops.dwconv3x3(a, wtb, F_, 8, 8)))
ar = a.clone().requires_grad_(); wr = wtb.clone().requires_grad_()
o2 = ops.dwconv3x3(ar, wr, F_, 8, 8)
go = g.view(F_, P, HID)
report("depthwise 3x3 bwd (dgrad + wgrad)", 4 * nb, timeit(lambda: torch.autograd.grad(o2, [ar, wr], go, retain_graph=True)))
x = torch.randn(R, C, device=dev); lw = torch.randn(C, device=dev); lb = torch.randn(C, device=dev)
nx = x.numel() * 4
report("LayerNorm(512) fwd", 2 * nx, timeit(lambda: ops.layernorm(x, lw, lb)))
xr = x.clone().requires_grad_()
y = ops.layernorm(xr, lw, lb)
gy = torch.randn_like(y)
report("LayerNorm(512) bwd", 3 * nx, timeit(lambda: torch.autograd.grad(y, xr, gy, retain_graph=True)))
# attention cores at the c1 size: spatial 4x4 windows over 320 frames; temporal T=10 over 32 clips
from npvp_amd.ops import AttnCfg
qk = torch.randn(R, 2 * C, device=dev, requires_grad=True); vv = torch.randn(R, C, device=dev, requires_grad=True)
cfg_s = AttnCfg(0, F_, 64, 8, 4, 0, 0, 8, 0, 0.1)
report("attention spatial fwd (q,k,v read + o write)", 4 * nx, timeit(lambda: ops.attn_packed(qk.detach(), vv.detach(), cfg_s)))
ya = ops.attn_packed(qk, vv, cfg_s); ga = torch.randn_like(ya)
report("attention spatial bwd (7 tensor passes)", 7 * nx, timeit(lambda: torch.autograd.grad(ya, [qk, vv], ga, retain_graph=True)))
T_ = 10
cfg_t = AttnCfg(1, F_ // T_, 64, 8, 0, T_, T_, 8, 1, 0.1)
report("attention temporal fwd (T=10, masked)", 4 * nx, timeit(lambda: ops.attn_packed(qk.detach(), vv.detach(), cfg_t)))
yt = ops.attn_packed(qk, vv, cfg_t)
report("attention temporal bwd (T=10, masked)", 7 * nx, timeit(lambda: torch.autograd.grad(yt, [qk, vv], ga, retain_graph=True)))
T2_ = 28
F2 = (F_ // T2_) * T2_ if F_ >= T2_ else T2_
R2 = F2 * 64
qk2 = torch.randn(R2, 2 * C, device=dev, requires_grad=True); vv2 = torch.randn(R2, C, device=dev, requires_grad=True)
cfg_t2 = AttnCfg(1, F2 // T2_, 64, 8, 0, T2_, T2_, 8, 0, 0.1)
nx2 = R2 * C * 4
report("attention temporal fwd (T=28)", 4 * nx2, timeit(lambda: ops.attn_packed(qk2.detach(), vv2.detach(), cfg_t2)))
yt2 = ops.attn_packed(qk2, vv2, cfg_t2); ga2 = torch.randn_like(yt2)
report("attention temporal bwd (T=28)", 7 * nx2, timeit(lambda: torch.autograd.grad(yt2, [qk2, vv2], ga2, retain_graph=True)))
