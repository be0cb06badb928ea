"""CPU oracle of the evaluation metrics (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Restates ref/utils/metrics.py in plain torch on whatever device the inputs live on (CPU in the tests):
  psnr_per_image / mse_per_image   ref :12-30 / :32-43
  ssim_per_image                   ref :46-108 (2-D Gaussian window = outer product of the normalised 1-D Gaussian,
                                   sigma 1.5; five zero-padded grouped convolutions; C1 = 0.01^2, C2 = 0.03^2)
  pred_ave_metrics                 ref :110-140
Pinned by tests/golden/metrics.npz, which tests/golden/make_golden.py writes from the reference's own functions.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def psnr_per_image(x, y, data_range=1.0):
    d = x / float(data_range) - y / float(data_range)
    return -10.0 * torch.log10((d * d).mean(dim=(1, 2, 3)) + 1e-8)


def mse_per_image(x, y):
    return ((x - y) ** 2).sum(dim=(1, 2, 3))


def gaussian_taps(window_size, sigma=1.5):
    g = torch.tensor([math.exp(-(i - window_size // 2) ** 2 / float(2 * sigma ** 2)) for i in range(window_size)], dtype=torch.float32)
    return g / g.sum()


def ssim_per_image(a, b, window_size=11):
    C = a.shape[1]
    g = gaussian_taps(window_size).to(a)
    win = torch.outer(g, g)[None, None].expand(C, 1, window_size, window_size).contiguous()
    blur = lambda t: F.conv2d(t, win, padding=window_size // 2, groups=C)
    ma, mb = blur(a), blur(b)
    va, vb, cab = blur(a * a) - ma * ma, blur(b * b) - mb * mb, blur(a * b) - ma * mb
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    smap = ((2 * ma * mb + c1) * (2 * cab + c2)) / ((ma * ma + mb * mb + c1) * (va + vb + c2))
    return smap.mean(dim=(1, 2, 3))


def pred_ave_metrics(model, data_loader, metric_func, renorm_transform, num_future_frames, device="cpu"):
    model = model.eval()
    tot, n = np.zeros(num_future_frames), 0
    with torch.no_grad():
        for past, fut in data_loader:
            past, fut = past.to(device), fut.to(device)
            pred = model(past, fut, None)[0]
            for t in range(num_future_frames):
                tot[t] += float(metric_func(renorm_transform(pred[:, t]), renorm_transform(fut[:, t]))) * pred.shape[0]
            n += pred.shape[0]
    return tot / n
