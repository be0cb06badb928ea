"""Single-shape traffic probe for the forward GEMM variants (plain / frame-statistics epilogue):
    rocprofv3 --pmc WRITE_SIZE -d <dir> -o g -- python3 tools/gemm_traffic_probe.py      (and again with FETCH_SIZE)
then tools/rocpd_pmc.py <db> --filter npvp::gemm."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npvp_amd import ops
dev = "cuda:0"
R = 114688
torch.manual_seed(0)
for N, K in [(2048, 512), (512, 2048)]:
    x = torch.randn(R, K, device=dev); w = torch.nn.Parameter(torch.randn(N, K, device=dev) / K ** 0.5); b = torch.randn(N, device=dev)
    for _ in range(3):
        ops.linear_fwd(x, w, b)                                  # plain epilogue
    for _ in range(3):
        part = torch.empty((R // 64) * (N // 64) * 2, device=dev)
        ops.linear_fwd(x, w, b, rowstats=part)                   # frame-statistics epilogue
torch.cuda.synchronize()
