"""Workload for the SQ-counter passes of the attention backward kernels:
    rocprofv3 --pmc <counters> --kernel-trace -d <dir> -o attn -- python tools/attn_pmc.py
temporal attention backward at the c2 decoder shapes (64 clips x 64 pixels x 8 heads; T = 28 self, 28 x 2 enc-dec)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npvp_amd import ops
from npvp_amd.ops import AttnCfg
dev = "cuda:0"
N, P, C = 64, 64, 512
torch.manual_seed(0)
for Tq, Tk in [(28, 28), (28, 2)]:
    q = torch.randn(N * Tq * P, C, device=dev, requires_grad=True); k = torch.randn(N * Tk * P, C, device=dev, requires_grad=True)
    v = torch.randn(N * Tk * P, C, device=dev, requires_grad=True)
    cfg = AttnCfg(1, N, P, 8, 0, Tq, Tk, 8, 0, 0.1)
    y = ops.attn(q, k, v, cfg); go = torch.randn_like(y)
    for _ in range(3):
        torch.autograd.grad(y, [q, k, v], go, retain_graph=True)
torch.cuda.synchronize()
