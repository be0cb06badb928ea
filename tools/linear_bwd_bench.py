"""dgrad + weight gradient of one linear layer: two launches on one stream against ONE launch (npvp_linear_bwd_f16), on the
shard-sized shapes.  Usage: python tools/linear_bwd_bench.py [--rows 8192 2048] [--iters 50]"""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import npvp_amd
from npvp_amd import ops
from npvp_amd.trainer import FlatBuffers

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, nargs="*", default=[8192, 4096, 2048])
ap.add_argument("--iters", type=int, default=50)
args = ap.parse_args()
dev = torch.device("cuda:0")
ops.set_gemm_precision("f16x3")
ops.WgradStream.enabled = False          # one stream: what a captured step sees
shapes = [(512, 512), (1024, 512), (2048, 512), (512, 2048), (512, 1024)]


def timeit(fn):
    for _ in range(3):
        fn()
    ops.ReduceQueue.finish()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    ops.ReduceQueue.finish()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters * 1e3


for R in args.rows:
    tot = [0.0, 0.0]
    for N, K in shapes:
        lin = torch.nn.Linear(K, N).to(dev)
        fb = FlatBuffers(lin)
        w, b = lin.weight, lin.bias
        sk = ops._wb_sink(w, b)
        x = torch.randn(R, K, device=dev); dy = torch.randn(R, N, device=dev)
        ops.amax_of(x); ops.amax_of(dy)
        res = []
        for fused in (False, True):
            ops.FusedLinearBwd.enabled = fused
            res.append(timeit(lambda: ops.linear_bwd(dy, x, w, b, sk)))
        tot[0] += res[0]; tot[1] += res[1]
        fl = 4.0 * R * N * K
        print(f"R={R:6d} N={N:5d} K={K:5d}  two launches {res[0]:7.1f} us ({fl/res[0]/1e6:6.1f} TF)   one launch {res[1]:7.1f} us ({fl/res[1]/1e6:6.1f} TF)"
              f"   taken: {ops.FusedLinearBwd.takes(R, N, K)}", flush=True)
    print(f"R={R:6d} all shapes: two launches {tot[0]:.1f} us, one launch {tot[1]:.1f} us", flush=True)
