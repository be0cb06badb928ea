"""PosFeatFuser backward at a workload's decoder size: the fused apply (batch loop in the thread, d beta / d gamma in registers)
against the apply + two reductions (the library's route for shapes npvp_posfuse_bwd_fused declines).  python tools/posfuse_bench.py [N] [T]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from npvp_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 28
dev = "cuda:0"
PF = 64 * 512
x = torch.randn(N * T, 64, 512, device=dev)
dy = torch.randn_like(x)
add = torch.randn(N, 64, 512, device=dev)
beta = torch.randn(T, 64, 512, device=dev)
gamma = torch.randn(T, 64, 512, device=dev) * 0.1
y, st = ops._raw_posfuse_fwd(x, add, beta, gamma, N, T)


def run():
    return ops._raw_posfuse_bwd(dy, x, add, beta.shape, gamma, st, N, T, True)


for _ in range(3):
    out = run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"N={N} T={T} fused={ops.lib().npvp_posfuse_bwd_fused(N, T, PF)}: {us:.1f} us per backward (stats + apply + reductions), "
      f"{x.numel() * 4 / 1e6:.0f} MB per tensor; checksum {float(out[2].double().sum()):.6e} {float(out[3].double().sum()):.6e} {float(out[0].double().sum()):.6e}")
