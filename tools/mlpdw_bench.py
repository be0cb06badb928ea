"""MlpDWBN forward + backward at a workload's decoder size (frames of 64 tokens, C = 512, hidden = 2048), stand-alone.
python tools/mlpdw_bench.py [frames]   (NPVP_WGRAD_STREAM=0 for one stream)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npvp_amd import ops

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1792
dev = "cuda:0"
C, hid, P = 512, 2048, 64
R = frames * P
torch.manual_seed(0)
g = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).requires_grad_()
x = g(R, C); res = g(R, C)
w1, b1 = g(hid, C, sc=C ** -0.5), g(hid, sc=0.1)
n1w, n1b = (1 + 0.1 * torch.randn(P * hid, device=dev)).requires_grad_(), g(P * hid, sc=0.1)
dww, dwb = g(hid, 1, 3, 3, sc=0.3), g(hid, sc=0.1)
n2w, n2b = (1 + 0.1 * torch.randn(P * hid, device=dev)).requires_grad_(), g(P * hid, sc=0.1)
w2, b2 = g(C, hid, sc=hid ** -0.5), g(C, sc=0.1)
n3w, n3b = (1 + 0.1 * torch.randn(P * C, device=dev)).requires_grad_(), g(P * C, sc=0.1)
cot = torch.randn(R, C, device=dev)
ops.rng.manual_seed(1, torch.device(dev))


def run():
    ops.rng.begin_step(torch.device(dev))
    y = ops.mlpdwbn(x, res, w1, b1, n1w, n1b, dww, dwb, n2w, n2b, w2, b2, n3w, n3b, frames, 28, 0.1, 0.1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr = torch.autograd.grad(y, [x, res, w1, b1, n1w, n1b, dww, dwb, n2w, n2b, w2, b2, n3w, n3b], cot)
    e1.record()
    return e0, e1, gr


for _ in range(3):
    run()
torch.cuda.synchronize()
ts = []
for _ in range(8):
    e0, e1, gr = run()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
print(f"frames={frames}: backward {ts[len(ts) // 2] * 1e3:.0f} us (median of 8); "
      f"checks dx {float(gr[0].double().abs().sum()):.6e} dn2w {float(gr[8].double().abs().sum()):.6e} ddww {float(gr[6].double().abs().sum()):.6e} "
      f"dn1w {float(gr[4].double().abs().sum()):.6e}")
