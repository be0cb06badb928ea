"""Workload for the SQ-counter passes of the GEMM kernels (profiles/r02_pmc_gemm.md):
    rocprofv3 --pmc <counters> --kernel-trace -d <dir> -o gemm -- python tools/gemm_pmc.py
runs each of forward / dgrad / wgrad a few times on the c2 decoder shapes (R = 114688 token rows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npvp_amd import ops
dev = "cuda:0"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 114688
torch.manual_seed(0)
for N, K in [(512, 512), (2048, 512), (512, 2048), (1024, 512)]:
    x = torch.randn(R, K, device=dev); w = torch.nn.Parameter(torch.randn(N, K, device=dev) / K ** 0.5)
    dy = torch.randn(R, N, device=dev); b = torch.randn(N, device=dev)
    for _ in range(3):
        ops.linear_fwd(x, w, b); ops.linear_dgrad(dy, w); ops.linear_wgrad(dy, x)
torch.cuda.synchronize()
