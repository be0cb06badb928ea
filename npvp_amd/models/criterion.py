"""Stage-2 loss terms (ref/models/criterion.py:99-121, 341-354).  On the device the scalar reductions are the library's
fixed-order sums (ops.l1_mean / ops.sum_all): deterministic, and a captured training step stays free of the memset nodes torch's
multi-block reductions bring (profiles/r06_graph_alloc_hazard.txt).  Host tensors take the stock torch formula."""
import torch
import torch.nn as nn
from .. import ops


class L1Loss(nn.Module):
    def __init__(self, norm_dim=None, lam=1.0):
        super().__init__()
        assert norm_dim is None, "norm_dim is never set on the Stage-2 path"
        self.norm_dim, self.lam = norm_dim, lam

    def __call__(self, gt, pred):
        # (argument names as in the reference: every caller passes the prediction first, and only that operand takes a gradient)
        if gt.is_cuda and gt.dtype == torch.float32 and pred.dtype == torch.float32 and not pred.requires_grad:
            return ops.l1_mean(gt, pred, self.lam)
        return torch.abs(pred - gt).mean() * self.lam


class Div_KL(nn.Module):
    def __init__(self, beta):
        super().__init__()
        self.beta = beta

    def forward(self, mu1, logvar1, mu2, logvar2):
        N = mu1.shape[0]
        sigma1, sigma2 = logvar1.mul(0.5).exp(), logvar2.mul(0.5).exp()
        kld = torch.log(sigma2 / sigma1) + (torch.exp(logvar1) + (mu1 - mu2) ** 2) / (2 * torch.exp(logvar2)) - 1 / 2
        total = ops.sum_all(kld) if kld.is_cuda and kld.dtype == torch.float32 else kld.sum()
        return self.beta * total / N
