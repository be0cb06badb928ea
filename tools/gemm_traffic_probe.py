"""Per-shape HBM-side traffic of the forward / dgrad GEMM kernels at the c2 decoder size (R = 114 688 token rows):
    rocprofv3 --pmc FETCH_SIZE -d <dir> -o g -- python3 tools/gemm_traffic_probe.py      (and again with WRITE_SIZE)
then tools/rocpd_pmc.py <fetch.db> <write.db> --filter npvp::gemm.  Launch order (3 launches each, grouped by workgroup count
in the summary): for (N, K) in the list: forward plain, forward with the frame-statistics epilogue (N % 128 == 0), dgrad."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npvp_amd import ops
dev = "cuda:0"
R = 114688
torch.manual_seed(0)
only = sys.argv[1:]
for N, K in [(512, 512), (1024, 512), (2048, 512), (512, 2048), (512, 1024)]:
    if only and f"{N}x{K}" not in only:
        continue
    x = torch.randn(R, K, device=dev); w = torch.nn.Parameter(torch.randn(N, K, device=dev) / K ** 0.5); b = torch.randn(N, device=dev)
    dy = torch.randn(R, N, device=dev)
    for _ in range(3):
        ops.linear_fwd(x, w, b)
    for _ in range(3):
        ops.linear_dgrad(dy, w)
torch.cuda.synchronize()
