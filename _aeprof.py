import sys, os
sys.path.insert(0, '/root/repo')
import torch, npvp_amd
dev = "cuda:0"
AE = {"ngf": 64, "n_downsampling": 3, "num_res_blocks": 2, "out_layer": "Tanh", "learn_3d": False}
enc, dec = npvp_amd.build_frozen_autoencoder(AE, 1)
enc, dec = npvp_amd.to_device_layout(enc, dec, dev)
x = torch.rand(32, 20, 1, 64, 64, device=dev)
f = torch.rand(32, 10, 512, 8, 8, device=dev, requires_grad=True)
for _ in range(4):
    with torch.no_grad():
        enc(x)
    y = dec(f); torch.autograd.grad(y.sum(), f)
torch.cuda.synchronize()
