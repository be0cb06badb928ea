"""Timing of the fused MlpDWBN middle kernels at a given frame count (default: c2 decoder, 1792 frames)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npvp_amd import ops
from npvp_amd._lib import lib, check
dev = "cuda:0"
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1792
Ch = 2048
L = lib(); P = ops._p; pt = ops._ptr
h1 = torch.randn(frames * 64, Ch, device=dev); dh2 = torch.randn_like(h1)
mean = torch.zeros(frames, device=dev); rstd = torch.ones(frames, device=dev)
w1n = torch.randn(64, Ch, device=dev) * 0.1 + 1; b1n = torch.randn(64, Ch, device=dev) * 0.1
wtb = torch.randn(10, Ch, device=dev) * 0.3
h2 = torch.empty_like(h1); da1 = torch.empty_like(h1); m2 = torch.empty(frames, device=dev); r2 = torch.empty(frames, device=dev)
ws, wsn = ops._ws(frames * (Ch // 512) * 8, torch.device(dev))
dwtb = torch.empty(10, Ch, device=dev); psum = torch.empty(frames * (Ch // 256) * 2, device=dev)
ws2, wsn2 = ops._ws(L.npvp_mlpdw_mid_bwd_workspace_bytes(frames, Ch), torch.device(dev))
st = ops._stream()
def fwd():
    check(L.npvp_mlpdw_mid_fwd(pt(h1), pt(mean), pt(rstd), pt(w1n), pt(b1n), pt(wtb), pt(wtb[9]), pt(h2), pt(m2), pt(r2), frames, 8, 8, Ch, 1e-5, pt(ws), wsn, st), "f")
def bwd():
    check(L.npvp_mlpdw_mid_bwd(pt(dh2), pt(h1), pt(mean), pt(rstd), pt(w1n), pt(b1n), pt(wtb), pt(da1), pt(dwtb), pt(psum), frames, 8, 8, Ch, 0, pt(ws2), wsn2, st), "b")
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
gb = h1.numel() * 4 / 1e9
tf, tb = timeit(fwd), timeit(bwd)
print(f"frames {frames}: mid_fwd {tf:.1f} us ({2*gb/tf*1e6/1e3:.2f} TB/s of 2 passes)   mid_bwd {tb:.1f} us ({3*gb/tb*1e6/1e3:.2f} TB/s of 3 passes)")
