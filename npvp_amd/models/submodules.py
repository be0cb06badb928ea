"""HIP-backed counterparts of ref/models/submodules.py:258-454 (predictor half): same class names,
constructor signatures, attribute names and state-dict keys; arithmetic through npvp_amd.ops.
"""
import math

import torch
import torch.nn as nn

from .. import ops


class CoorGenerator(nn.Module):
    """ref/models/submodules.py:329-366.  Runs once at construction (host side, tiny)."""

    def __init__(self, max_H, max_W, max_T):
        super().__init__()
        self.max_H, self.max_W, self.max_T = max_H, max_W, max_T

    def forward(self, t_list, h_list, w_list):
        assert torch.max(h_list) <= self.max_H and torch.min(h_list) >= 0., "Invalid H coordinates"
        assert torch.max(w_list) <= self.max_W and torch.min(w_list) >= 0., "Invalid W coordinates"
        assert torch.max(t_list) <= self.max_T and torch.min(t_list) >= 0., "Invalid T coordinates"
        t = (t_list / self.max_T).view(-1, 1, 1)
        h = (h_list / self.max_H).view(1, -1, 1)
        w = (w_list / self.max_W).view(1, 1, -1)
        T, H, W = t.shape[0], h.shape[1], w.shape[2]
        coor = torch.stack([t.expand(T, H, W), h.expand(T, H, W), w.expand(T, H, W)], dim=-1)
        return coor.reshape(T * H * W, 3)


class NRMLP(nn.Module):
    """ref/models/submodules.py:258-327.  <= T*64 rows per call: the four Linear layers run on the
    MFMA GEMM kernel, the 3-wide Fourier projection / cos / sin / ReLU glue stays in stock torch ops
    (negligible, input independent)."""

    def __init__(self, out_channels, dim_x=3, d_model=256, MLP_layers=4, scale=10, fix_B=False, fuse_method='SPADE'):
        super().__init__()
        self.scale, self.dim_x, self.out_channels = scale, dim_x, out_channels
        self.MLP_layers, self.d_model, self.fix_B = MLP_layers, d_model, fix_B
        B = torch.normal(mean=0, std=1.0, size=(d_model, dim_x)) * scale
        if fix_B:
            self.register_buffer('B', B)
        else:
            self.B = nn.Parameter(B, requires_grad=True)
        layers = [nn.Linear(2 * d_model, d_model), nn.ReLU()]
        for _ in range(MLP_layers - 2):
            layers += [nn.Linear(d_model, d_model), nn.ReLU()]
        self.MLP = nn.Sequential(*layers)
        self.fuse_method = fuse_method
        self.mlp_beta = nn.Linear(d_model, out_channels)
        if fuse_method == 'SPADE':
            self.mlp_gamma = nn.Linear(d_model, out_channels)

    def gaussian_mapping(self, x):
        proj = (2. * float(math.pi) * x) @ self.B.T
        return torch.cat([torch.cos(proj), torch.sin(proj)], dim=-1)

    def forward(self, x):
        x = self.gaussian_mapping(x)
        for m in self.MLP:
            x = ops.linear(x, m.weight, m.bias) if isinstance(m, nn.Linear) else torch.relu(x)
        beta = ops.linear(x, self.mlp_beta.weight, self.mlp_beta.bias)
        if self.fuse_method == 'SPADE':
            gamma = ops.linear(x, self.mlp_gamma.weight, self.mlp_gamma.bias)
        else:
            gamma = torch.zeros_like(beta)     # as the reference (ref :309-312); Predictor drops it (1 + 0 is exact)
        return beta, gamma


class PosFeatFuser(nn.Module):
    """ref/models/submodules.py:412-454, param_free_norm_type 'layer' (every shipped config)."""

    def __init__(self, x_channels, param_free_norm_type='layer'):
        super().__init__()
        if param_free_norm_type != 'layer':
            raise NotImplementedError(
                f"param_free_norm_type={param_free_norm_type!r}: only 'layer' (GroupNorm(1,C)) has a HIP kernel; "
                "all reference configs use 'layer'")
        self.norm_type = param_free_norm_type

    def forward(self, x, pos_beta, pos_gamma, add=None):
        """x (N,T,H,W,C); pos_* (T*H*W, C) (pos_gamma may be None = zeros); add (N,H,W,C) optional."""
        N, T = x.shape[0], x.shape[1]
        return ops.posfuse(x, add, pos_beta, pos_gamma, N, T).view(x.shape)


class EventEncoder(nn.Module):
    """ref/models/submodules.py:368-410.  (N,512,8,8) in, 2 calls per step: stock PyTorch-ROCm convs /
    BatchNorm (SURVEY K11 - not a kernel target).  Under data parallelism the BatchNorm layers are
    swapped for npvp_amd.dp.SyncBatchNorm2d (the reference trains with sync_batchnorm=True)."""

    def __init__(self, in_channels, hidden_channels, n_layers, stochastic):
        super().__init__()
        self.stochastic, self.n_layers = stochastic, n_layers
        self.conv1 = nn.Sequential(
            nn.Conv2d(in_channels, in_channels, 3, 1, 1, bias=False, groups=in_channels),
            nn.BatchNorm2d(in_channels), nn.ReLU(True))
        self.conv2 = nn.Sequential(
            nn.Conv2d(in_channels, hidden_channels, 3, 1, 1, bias=False),
            nn.BatchNorm2d(hidden_channels), nn.ReLU(True))
        for i in range(n_layers):
            setattr(self, f'MLP_{i}', nn.Sequential(
                nn.Conv2d(hidden_channels, hidden_channels, 1, 1, bias=False),
                nn.BatchNorm2d(hidden_channels), nn.ReLU(True)))
        self.mu_net = nn.Conv2d(hidden_channels, in_channels, 1, 1, bias=True)
        if stochastic:
            self.logvar_net = nn.Conv2d(hidden_channels, in_channels, 1, 1, bias=True)
        self.eps_fn = None   # test hook: callable(shape) -> eps

    def forward(self, x):
        x = self.conv2(self.conv1(x))
        for i in range(self.n_layers):
            x = getattr(self, f'MLP_{i}')(x)
        mu = self.mu_net(x)
        if self.stochastic:
            logvar = self.logvar_net(x)
            return self.reparameterize(mu, logvar), mu, logvar
        return mu

    def reparameterize(self, mu, logvar):
        eps = self.eps_fn(mu.shape) if self.eps_fn is not None else torch.randn(mu.shape, device=mu.device)
        return mu + torch.exp(0.5 * logvar) * eps
