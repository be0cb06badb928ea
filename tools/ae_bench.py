"""Frozen autoencoder (stock PyTorch-ROCm) timing at the c1 full-step size: encoder on 32x20 frames (no grad), decoder
forward + input-gradient on 32x10 frames; contiguous vs channels_last vs the fused product path.  Usage: python tools/ae_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import npvp_amd

dev = "cuda:0"
AE = {"ngf": 64, "n_downsampling": 3, "num_res_blocks": 2, "out_layer": "Tanh", "learn_3d": False}


def timeit(fn, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for cl in (False, True, "fused"):
    enc, dec = npvp_amd.build_frozen_autoencoder(AE, 1)
    if cl == "fused":          # the product path: channels_last encoder, folded BatchNorm, csrc/ae.hip epilogues
        enc, dec = npvp_amd.to_device_layout(enc, dec, dev)
    else:
        enc, dec = enc.to(dev), dec.to(dev)
        if cl:
            enc, dec = enc.to(memory_format=torch.channels_last), dec.to(memory_format=torch.channels_last)
    x = torch.rand(32, 20, 1, 64, 64, device=dev)
    f = torch.rand(32, 10, 512, 8, 8, device=dev, requires_grad=True)

    def e():
        with torch.no_grad():
            return enc(x)

    def d():
        y = dec(f)
        (g,) = torch.autograd.grad(y.sum(), f)
        return g
    print(f"channels_last={cl}: encoder {timeit(e):.2f} ms, decoder fwd+input-grad {timeit(d):.2f} ms", flush=True)
