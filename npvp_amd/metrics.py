"""Evaluation metrics on device: drop-in for ref/utils/metrics.py:12-140 (`PSNR`, `MSEScore`, `SSIM`,
`pred_ave_metrics`) - same names, arguments and return types - with the reductions and the five Gaussian-filtered
maps of SSIM done by libnpvp_hip.so (csrc/metrics.hip: every image pair is read once).  Inputs must be fp32 tensors
on an MI355X; there is no CPU path (RuntimeError), as everywhere in this package.
"""
from math import exp

import numpy as np
import torch

from . import ops
from ._lib import lib


def _pair(x, y):
    if x.shape != y.shape or x.dim() != 4:
        raise ValueError(f"metrics: two (N, C, H, W) tensors of equal shape are needed, got {tuple(x.shape)} and {tuple(y.shape)}")
    if not (x.is_cuda and y.is_cuda):
        raise RuntimeError("npvp_amd.metrics runs on the GPU only (no CPU fallback): move the frames to the device")
    return x.detach().float().contiguous(), y.detach().float().contiguous()


def _sqdiff(x, y, data_range, scale):
    x, y = _pair(x, y)
    N, per = x.shape[0], x[0].numel()
    L = lib()
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    ws, wsn = ops._ws(L.npvp_sqdiff_workspace_bytes(N, per), x.device)
    ops.check(L.npvp_sqdiff_per_image(ops._ptr(x), ops._ptr(y), N, per, float(data_range), float(scale), ops._ptr(out),
                                       ops._ptr(ws), wsn, ops._stream()), "npvp_sqdiff_per_image")
    return out


def PSNR(x, y, data_range=1.0, mean_flag=True):
    """average (or per-image) PSNR of two batches of (N, C, H, W) images   ref/utils/metrics.py:12-30"""
    mse = _sqdiff(x, y, data_range, 1.0 / x[0].numel())
    score = -10 * torch.log10(mse + 1e-8)
    return torch.mean(score).item() if mean_flag else score


def MSEScore(x, y, mean_flag=True):
    """per-image SUM of squared differences (ref/utils/metrics.py:32-43), averaged over the batch if mean_flag"""
    mse = _sqdiff(x, y, 1.0, 1.0)
    return torch.mean(mse).item() if mean_flag else mse


class SSIM(torch.nn.Module):
    """ref/utils/metrics.py:46-108: Gaussian-window SSIM (sigma 1.5, zero padding), C1 = 0.01^2, C2 = 0.03^2."""

    def __init__(self, window_size=11):
        super().__init__()
        self.window_size = window_size
        self.channel = 1
        self.window = self.create_window(window_size, self.channel)
        self.__name__ = 'SSIM'

    def gaussian(self, window_size, sigma):
        gauss = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
        return gauss / gauss.sum()

    def create_window(self, window_size, channel):
        """the reference's (channel, 1, ws, ws) grouped-conv window; the kernel applies its two 1-D factors separably"""
        g = self.gaussian(window_size, 1.5).unsqueeze(1)
        return g.mm(g.t()).float().unsqueeze(0).unsqueeze(0).expand(channel, 1, window_size, window_size).contiguous()

    def forward(self, img1, img2, mean_flag=True):
        img1, img2 = _pair(img1, img2)
        N, C, H, W = img1.shape
        self.channel = C
        taps = np.ascontiguousarray(self.gaussian(self.window_size, 1.5).numpy(), dtype=np.float32)
        L = lib()
        out = torch.empty(N, dtype=torch.float32, device=img1.device)
        ws, wsn = ops._ws(L.npvp_ssim_workspace_bytes(N, C, H, W), img1.device)
        ops.check(L.npvp_ssim_per_image(ops._ptr(img1), ops._ptr(img2), N, C, H, W, taps.ctypes.data, self.window_size,
                                         ops._ptr(out), ops._ptr(ws), wsn, ops._stream()), "npvp_ssim_per_image")
        return out.mean() if mean_flag else out


def pred_ave_metrics(model, data_loader, metric_func, renorm_transform, num_future_frames, ckpt=None, device='cuda:0'):
    """Per-time-step average of `metric_func` over a data loader (ref/utils/metrics.py:110-140).  `model(past, future,
    None)[0]` must give the predicted frames (N, Tp, C, H, W) as in the reference; `ckpt`, if given, is a Lightning
    checkpoint loaded with npvp_amd.load_lightning_checkpoint into `model.predictor` (or `model`)."""
    if ckpt is not None:
        from .trainer import load_lightning_checkpoint
        load_lightning_checkpoint(ckpt, getattr(model, "predictor", model))
    model = model.eval()
    ave_metric = np.zeros(num_future_frames)
    sample_num = 0
    with torch.no_grad():
        for past_frames, future_frames in data_loader:
            past_frames, future_frames = past_frames.to(device), future_frames.to(device)
            pred = model(past_frames, future_frames, None)[0]
            for i in range(num_future_frames):
                pred_t, gt_t = pred[:, i, ...], future_frames[:, i, ...]
                m = metric_func(renorm_transform(pred_t), renorm_transform(gt_t))
                ave_metric[i] += float(m) * pred_t.shape[0]
            sample_num += pred.shape[0]
    return ave_metric / sample_num
