"""HIP-backed counterparts of ref/models/submodules.py:258-454 (predictor half): same class names,
constructor signatures, attribute names and state-dict keys; arithmetic through npvp_amd.ops.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

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
    """ref/models/submodules.py:412-454: param_free_norm_type 'layer' (GroupNorm(1, C): every shipped config; the sub-layer nodes
    of the blocks call its kernels directly) or 'instance' (InstanceNorm2d: statistics per frame and channel)."""

    def __init__(self, x_channels, param_free_norm_type='layer'):
        super().__init__()
        if param_free_norm_type not in ('layer', 'instance'):
            raise ValueError('%s is not a supported param-free norm type' % param_free_norm_type)        # ref :432-433
        self.norm_type = param_free_norm_type

    def forward(self, x, pos_beta, pos_gamma, add=None):
        """x (N,T,H,W,C); pos_* (T*H*W, C) (pos_gamma may be None = zeros); add (N,H,W,C) optional."""
        N, T = x.shape[0], x.shape[1]
        if self.norm_type == 'instance':
            return ops.posfuse_instance(x, add, pos_beta, pos_gamma, N, T).view(x.shape)
        return ops.posfuse(x, add, pos_beta, pos_gamma, N, T).view(x.shape)


def _bn_rows(x2, bn):
    """BatchNorm2d semantics on channels-last rows x2 [R, C] (statistics over the R = N*H*W rows per channel).
    Tiny tensors ((N*64) x 512): stock torch batch-norm kernels; under data parallelism `bn` has been swapped
    for npvp_amd.dp.SyncBatchNorm2d and the statistics span all ranks."""
    from ..dp import SyncBatchNorm2d, _SyncBNFn, active
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    if isinstance(bn, SyncBatchNorm2d) and bn.training and active():      # (more than one rank, or one rank with NPVP_DP_FORCE=1)
        from ..dp import syncbn_group
        return _SyncBNFn.apply(x2, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, True, syncbn_group())
    return F.batch_norm(x2, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, bn.momentum, bn.eps)


class EventEncoder(nn.Module):
    """ref/models/submodules.py:368-410.  Same modules / state-dict keys (conv1.0, conv1.1, conv2.0, ...), but the
    arithmetic runs channels-last on the canonical layout: the depthwise 3x3 is the MlpDWBN dwconv kernel, the dense
    3x3 is im2col + the MFMA GEMM, the 1x1 convs are GEMMs (MIOpen's fallback for these shapes is a naive
    direct-conv kernel that cost 10 % of a training step); BatchNorm + ReLU on the (N*64) x C rows stay stock torch
    ops.  Under data parallelism the BatchNorm layers are npvp_amd.dp.SyncBatchNorm2d (the reference
    trains with sync_batchnorm=True, ref/train_Predictor_lightning.py:41)."""

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
        self.eps_fn = None   # test hook: callable(shape (N,C,H,W)) -> eps

    def forward_canonical(self, x, H, W):
        """x [N, H*W, C] -> mu (and z, logvar) as [N, H*W, C]."""
        N, P, C = x.shape
        w1 = self.conv1[0].weight                                        # [C,1,3,3] depthwise, no bias
        wtb = torch.cat([ops._Transpose.apply(w1.reshape(1, C, 9)).reshape(9, C), torch.zeros(1, C, device=x.device)], 0)
        h = ops.dwconv3x3(x, wtb, N, H, W).reshape(N * P, C)
        h = torch.relu(_bn_rows(h, self.conv1[1]))
        w2 = self.conv2[0].weight                                        # [hid, C, 3, 3] -> tap-major [hid, 9*C]
        cols = ops.im2col3x3(h.view(N, P, C), N, H, W)
        h = ops.linear(cols, w2.permute(0, 2, 3, 1).reshape(w2.shape[0], 9 * C))
        h = torch.relu(_bn_rows(h, self.conv2[1]))
        for i in range(self.n_layers):
            m = getattr(self, f'MLP_{i}')
            h = torch.relu(_bn_rows(ops.linear(h, m[0].weight.flatten(1)), m[1]))
        mu = ops.linear(h, self.mu_net.weight.flatten(1), self.mu_net.bias).view(N, P, C)
        if not self.stochastic:
            return mu
        logvar = ops.linear(h, self.logvar_net.weight.flatten(1), self.logvar_net.bias).view(N, P, C)
        if self.eps_fn is not None:
            eps = ops.transpose(self.eps_fn((N, C, H, W)).reshape(N, C, P))
        else:
            eps = torch.randn(N, P, C, device=x.device)
        return mu + torch.exp(0.5 * logvar) * eps, mu, logvar

    def forward(self, x):
        """x (N,C,H,W) -> mu | (z, mu, logvar), each (N,C,H,W)   (reference signature)"""
        N, C, H, W = x.shape
        out = self.forward_canonical(ops._Transpose.apply(x.reshape(N, C, H * W)), H, W)
        back = lambda t: ops._Transpose.apply(t).view(N, C, H, W)
        return tuple(back(t) for t in out) if self.stochastic else back(out)

    def reparameterize(self, mu, logvar):
        eps = self.eps_fn(mu.shape) if self.eps_fn is not None else torch.randn(mu.shape, device=mu.device)
        return mu + torch.exp(0.5 * logvar) * eps
