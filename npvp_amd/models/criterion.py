"""Stage-2 loss terms (ref/models/criterion.py:99-121, 341-354).  Scalar reductions over the
predictor output: stock torch ops on the device."""
import torch
import torch.nn as nn


class L1Loss(nn.Module):
    def __init__(self, norm_dim=None, lam=1.0):
        super().__init__()
        assert norm_dim is None, "norm_dim is never set on the Stage-2 path"
        self.norm_dim, self.lam = norm_dim, lam

    def __call__(self, gt, pred):
        return torch.abs(pred - gt).mean() * self.lam


class Div_KL(nn.Module):
    def __init__(self, beta):
        super().__init__()
        self.beta = beta

    def forward(self, mu1, logvar1, mu2, logvar2):
        N = mu1.shape[0]
        sigma1, sigma2 = logvar1.mul(0.5).exp(), logvar2.mul(0.5).exp()
        kld = torch.log(sigma2 / sigma1) + (torch.exp(logvar1) + (mu1 - mu2) ** 2) / (2 * torch.exp(logvar2)) - 1 / 2
        return self.beta * kld.sum() / N
