"""CPU restatement of the NPVP Stage-2 predictor as nn.Modules with the
reference's class names, constructor signatures, attribute names and
state-dict keys, written against the canonical [F, P, C] layout of
oracle/ops.py.  TEST INFRASTRUCTURE - see oracle/__init__.py.
"""
import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

__all__ = [
    "CoorGenerator", "NRMLP", "PosFeatFuser", "EventEncoder", "MultiheadAttention",
    "SpatialLocalMultiheadAttention", "MlpDWBN", "VidHRFormerBlockEnc",
    "VidHRFormerEncoder", "VidHRFormerBlockDecNAR", "VidHRformerDecoderNAR",
    "Predictor", "L1Loss", "Div_KL", "predictor_train_step", "full_train_step", "predictor_val_step", "full_val_step", "build_predictor_from_cfg",
    "rand_context_batch_process",
]


class CoorGenerator(nn.Module):
    """ref/models/submodules.py:329-366: normalised (t, h, w) grid, rows in (t,h,w) order."""

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
    """ref/models/submodules.py:258-327: learnable-B Fourier features + ReLU MLP."""

    def __init__(self, out_channels, dim_x=3, d_model=256, MLP_layers=4, scale=10,
                 fix_B=False, fuse_method='SPADE'):
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
        x = self.MLP(self.gaussian_mapping(x))
        beta = self.mlp_beta(x)
        gamma = self.mlp_gamma(x) if self.fuse_method == 'SPADE' else torch.zeros_like(beta)
        return beta, gamma


class PosFeatFuser(nn.Module):
    """ref/models/submodules.py:412-454.  Parameter-free."""

    def __init__(self, x_channels, param_free_norm_type='layer'):
        super().__init__()
        if param_free_norm_type not in ('layer', 'instance'):
            raise ValueError('%s is not a supported param-free norm type' % param_free_norm_type)
        self.norm_type = param_free_norm_type

    def forward(self, x, pos_beta, pos_gamma, add=None):
        """x: (N,T,H,W,C) -> (N,T,H,W,C).  `add` (N,H,W,C) is an extension used by
        the decoder: fuse(x + add broadcast over T)."""
        N, T, H, W, C = x.shape
        y = ops.posfuse(x.reshape(N * T, H * W, C), T, pos_beta, pos_gamma,
                        None if add is None else add.reshape(N, H * W, C), self.norm_type)
        return y.reshape(N, T, H, W, C)


class EventEncoder(nn.Module):
    """ref/models/submodules.py:368-410."""

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
        self.eps_fn = None   # test hook: callable(shape) -> eps, replaces torch.randn

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


class _OutProj(nn.Module):
    def __init__(self, C):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(C, C))
        self.bias = nn.Parameter(torch.zeros(C))
        nn.init.xavier_uniform_(self.weight)


class MultiheadAttention(nn.Module):
    """Parameter layout of torch.nn.MultiheadAttention (in_proj_weight [3C,C],
    in_proj_bias [3C], out_proj.{weight,bias}); arithmetic = its slow path with
    key is not value, as used at ref/models/VidHRFormer.py:70,180,192,270."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = _OutProj(embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)

    def forward(self, xq, xk, xv, q_rows, k_rows, mask=None):
        """xq [Rq,C], xk/xv [Rk,C] canonical rows; returns [Rq,C]."""
        C = self.embed_dim
        w, b = self.in_proj_weight, self.in_proj_bias
        q = ops.linear(xq, w[0:C], b[0:C])
        k = ops.linear(xk, w[C:2 * C], b[C:2 * C])
        v = ops.linear(xv, w[2 * C:], b[2 * C:])
        o = ops.attn_core(q, k, v, q_rows, k_rows, self.num_heads, mask, self.dropout, self.training)
        return ops.linear(o, self.out_proj.weight, self.out_proj.bias)


class SpatialLocalMultiheadAttention(nn.Module):
    """ref/models/VidHRFormer.py:247-307: 16-token window MHA, q=k source `x`, separate value."""

    def __init__(self, embed_dim, num_heads, window_size=7, dropout=0.):
        super().__init__()
        self.dim, self.num_heads, self.window_size, self.dropout = embed_dim, num_heads, window_size, dropout
        self.attn = MultiheadAttention(embed_dim, num_heads, dropout=dropout)

    def forward(self, x, value=None):
        N, T, H, W, C = x.shape
        rows = ops.spatial_groups(N * T, H, W, self.window_size)
        xq = x.reshape(-1, C)
        xv = xq if value is None else value.reshape(-1, C)
        return self.attn(xq, xq, xv, rows, rows).reshape(N, T, H, W, C)


class MlpDWBN(nn.Module):
    """ref/models/VidHRFormer.py:326-392 (AR_model=True => LayerNorm((C,H,W)) variant)."""

    def __init__(self, encH, encW, in_features, hidden_features=None, out_features=None,
                 act_layer=nn.GELU, dw_act_layer=nn.GELU, drop=0.0, AR_model=True):
        super().__init__()
        assert AR_model, "only the LayerNorm variant is on the hot path"
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.encH, self.encW = encH, encW
        self.fc1 = nn.Conv2d(in_features, hidden_features, kernel_size=1)
        self.norm1 = nn.LayerNorm((hidden_features, encH, encW))
        self.dw3x3 = nn.Conv2d(hidden_features, hidden_features, 3, 1, 1, groups=hidden_features)
        self.norm2 = nn.LayerNorm((hidden_features, encH, encW))
        self.fc2 = nn.Conv2d(hidden_features, out_features, kernel_size=1)
        self.norm3 = nn.LayerNorm((out_features, encH, encW))
        self.drop = nn.Dropout(drop)
        self.out_features = out_features

    @staticmethod
    def _cl(p):   # (Ch,H,W) state-dict tensor -> [P, Ch] channels-last
        return p.permute(1, 2, 0).reshape(-1, p.shape[0])

    def forward(self, x):
        N, T, H, W, C = x.shape
        h = x.reshape(N * T, H * W, C)
        h = ops.linear(h, self.fc1.weight.flatten(1), self.fc1.bias)
        h = ops.gelu(ops.frame_ln(h, self._cl(self.norm1.weight), self._cl(self.norm1.bias)))
        h = ops.dwconv3x3(h, self.dw3x3.weight.squeeze(1), self.dw3x3.bias, H, W)
        h = ops.gelu(ops.frame_ln(h, self._cl(self.norm2.weight), self._cl(self.norm2.bias)))
        h = self.drop(h)
        h = ops.linear(h, self.fc2.weight.flatten(1), self.fc2.bias)
        h = ops.gelu(ops.frame_ln(h, self._cl(self.norm3.weight), self._cl(self.norm3.bias)))
        h = self.drop(h)
        return h.reshape(N, T, H, W, self.out_features)


def _drop_path(x, p, training, dim):
    """ref/models/VidHRFormer.py:513-525 applied along `dim` of a (N,T,H,W,C) tensor:
    dim=0 is the reference's per-sample case, dim=1 the per-time-step case that
    arises at ref/models/VidHRFormer.py:239 where the tensor is (T2, N*H*W, C)."""
    if p == 0.0 or not training:
        return x
    keep = 1 - p
    shape = [1] * x.ndim
    shape[dim] = x.shape[dim]
    r = (keep + torch.rand(shape, dtype=x.dtype, device=x.device)).floor_()
    return x.div(keep) * r


def _get_clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


class VidHRFormerBlockEnc(nn.Module):
    """ref/models/VidHRFormer.py:54-116."""

    def __init__(self, encH, encW, embed_dim, num_heads, window_size=7, dropout=0., drop_path=0.,
                 Spatial_FFN_hidden_ratio=4, dim_feedforward=1024):
        super().__init__()
        self.embed_dim, self.num_heads, self.window_size, self.dropout = embed_dim, num_heads, window_size, dropout
        self.drop_path_p = drop_path
        self.SLMHSA = SpatialLocalMultiheadAttention(embed_dim, num_heads, window_size, dropout)
        self.SpatialFFN = MlpDWBN(encH, encW, embed_dim, int(Spatial_FFN_hidden_ratio * embed_dim), embed_dim, drop=dropout)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.norm3 = nn.LayerNorm(embed_dim)
        self.temporal_MHSA = MultiheadAttention(embed_dim, num_heads, dropout=dropout)
        self.linear1 = nn.Linear(embed_dim, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, embed_dim)
        self.norm4 = nn.LayerNorm(embed_dim)

    def _ln(self, m, x):
        return ops.layernorm(x, m.weight, m.bias, m.eps)

    def forward(self, x, memory_pos, pos_fuser):
        N, T, H, W, C = x.shape
        P = H * W
        tr, dp = self.training, self.drop_path_p
        x1 = self._ln(self.norm1, x)
        x = x + _drop_path(self.SLMHSA(pos_fuser(x1, *memory_pos), value=x1), dp, tr, 0)
        x = x + _drop_path(self.SpatialFFN(self._ln(self.norm2, x)), dp, tr, 0)
        x1 = self._ln(self.norm3, x)
        temp = pos_fuser(x1, *memory_pos)
        rows = ops.temporal_groups(N, T, P)
        a = self.temporal_MHSA(temp.reshape(-1, C), temp.reshape(-1, C), x1.reshape(-1, C), rows, rows,
                               ops.encoder_temporal_mask(T))
        x = x + F.dropout(a.reshape(N, T, H, W, C), self.dropout, tr)
        x1 = self._ln(self.norm4, x)
        x1 = self.linear2(F.dropout(ops.gelu(self.linear1(x1)), self.dropout, tr))
        return x + F.dropout(x1, self.dropout, tr)


class VidHRFormerEncoder(nn.Module):
    """ref/models/VidHRFormer.py:11-52 (evt_token=False path only)."""

    def __init__(self, num_layers, enc_H, enc_W, d_model, num_heads, window_size=7, dropout=0.,
                 drop_path=0., Spatial_FFN_hidden_ratio=4, dim_feedforward=1024, norm=None, evt_token=False):
        super().__init__()
        assert not evt_token, "learn_evt_token branch is out of scope (ref Predictor.py:46 passes False)"
        self.layers = _get_clones(VidHRFormerBlockEnc(enc_H, enc_W, d_model, num_heads, window_size, dropout,
                                                      drop_path, Spatial_FFN_hidden_ratio, dim_feedforward), num_layers)
        self.num_layers, self.norm, self.evt_token = num_layers, norm, evt_token

    def forward(self, src, memory_pos, pos_fuser):
        out = src.permute(0, 1, 3, 4, 2).contiguous()
        for layer in self.layers:
            out = layer(out, memory_pos, pos_fuser)
        if self.norm is not None:
            out = ops.layernorm(out, self.norm.weight, self.norm.bias, self.norm.eps)
        return out.permute(0, 1, 4, 2, 3)


class VidHRFormerBlockDecNAR(nn.Module):
    """ref/models/VidHRFormer.py:163-245."""

    def __init__(self, encH, encW, embed_dim, num_heads, window_size=7, dropout=0., drop_path=0.,
                 Spatial_FFN_hidden_ratio=4, dim_feedforward=1024):
        super().__init__()
        self.embed_dim, self.num_heads, self.window_size, self.dropout = embed_dim, num_heads, window_size, dropout
        self.drop_path_p = drop_path
        hid = int(Spatial_FFN_hidden_ratio * embed_dim)
        self.SLMHSA = SpatialLocalMultiheadAttention(embed_dim, num_heads, window_size, dropout)
        self.SpatialFFN = MlpDWBN(encH, encW, embed_dim, hid, embed_dim, drop=dropout)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.norm3 = nn.LayerNorm(embed_dim)
        self.temporal_MHSA = MultiheadAttention(embed_dim, num_heads, dropout=dropout)
        self.linear1 = nn.Linear(embed_dim, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, embed_dim)
        self.norm4 = nn.LayerNorm(embed_dim)
        self.EncDecAttn = MultiheadAttention(embed_dim, num_heads, dropout=dropout)
        self.SpatialFFN1 = MlpDWBN(encH, encW, embed_dim, hid, embed_dim, drop=dropout)
        self.norm5 = nn.LayerNorm(embed_dim)
        self.norm6 = nn.LayerNorm(embed_dim)

    def _ln(self, m, x):
        return ops.layernorm(x, m.weight, m.bias, m.eps)

    def forward(self, tgt, query_evt, memory, memory_pos, tgt_pos, pos_fuser, fused_memory=None):
        """tgt (N,T2,H,W,C); query_evt (N,H,W,C) [= z, NOT repeated over T2];
        memory (N,T1,H,W,C).  `fused_memory` lets the caller hoist the
        layer-invariant pos_fuser(memory) of ref/models/VidHRFormer.py:232."""
        N, T2, H, W, C = tgt.shape
        T1, P = memory.shape[1], H * W
        tr, dp = self.training, self.drop_path_p
        t2 = self._ln(self.norm1, tgt)
        a = self.SLMHSA(pos_fuser(t2, *tgt_pos, add=query_evt), value=t2)
        x = tgt + _drop_path(a, dp, tr, 0)
        x = x + _drop_path(self.SpatialFFN(self._ln(self.norm2, x)), dp, tr, 0)
        x1 = self._ln(self.norm3, x)
        temp = pos_fuser(x1, *tgt_pos)
        rows = ops.temporal_groups(N, T2, P)
        a = self.temporal_MHSA(temp.reshape(-1, C), temp.reshape(-1, C), x1.reshape(-1, C), rows, rows)
        x = x + F.dropout(a.reshape(N, T2, H, W, C), self.dropout, tr)
        x1 = self._ln(self.norm4, x)
        x1 = self.linear2(F.dropout(ops.gelu(self.linear1(x1)), self.dropout, tr))
        x = x + F.dropout(x1, self.dropout, tr)
        x1 = self._ln(self.norm5, x)
        key = fused_memory if fused_memory is not None else pos_fuser(memory, *memory_pos)
        query = pos_fuser(x1, *tgt_pos, add=query_evt)
        a = self.EncDecAttn(query.reshape(-1, C), key.reshape(-1, C), memory.reshape(-1, C),
                            rows, ops.temporal_groups(N, T1, P))
        x = x + _drop_path(a.reshape(N, T2, H, W, C), dp, tr, 1)   # per TIME-STEP (VidHRFormer.py:239)
        x = x + _drop_path(self.SpatialFFN1(self._ln(self.norm6, x)), dp, tr, 0)
        return x


class VidHRformerDecoderNAR(nn.Module):
    """ref/models/VidHRFormer.py:118-161 (return_intermediate=False only)."""

    def __init__(self, num_layers, encH, encW, embed_dim, num_heads, window_size=7, dropout=0., drop_path=0.,
                 Spatial_FFN_hidden_ratio=4, dim_feedforward=1024, norm=None, return_intermediate=False):
        super().__init__()
        assert not return_intermediate, "return_intermediate is never used by the predictor"
        self.layers = _get_clones(VidHRFormerBlockDecNAR(encH, encW, embed_dim, num_heads, window_size, dropout,
                                                         drop_path, Spatial_FFN_hidden_ratio, dim_feedforward), num_layers)
        self.num_layers, self.norm, self.return_intermediate = num_layers, norm, return_intermediate

    def forward(self, query_evt, memory, memory_pos, tgt_pos, pos_fuser):
        """query_evt (N,T2,C,H,W) - every time-step identical (z repeated, ref
        Predictor.py:317,321,332); memory (N,T1,C,H,W)."""
        N, T2, C, H, W = query_evt.shape
        qe = query_evt[:, 0].permute(0, 2, 3, 1).contiguous()          # (N,H,W,C)
        memory = memory.permute(0, 1, 3, 4, 2).contiguous()
        out = torch.zeros(N, T2, H, W, C, dtype=memory.dtype)
        fused_memory = pos_fuser(memory, *memory_pos)
        for layer in self.layers:
            out = layer(out, qe, memory, memory_pos, tgt_pos, pos_fuser, fused_memory)
        if self.norm is not None:
            out = ops.layernorm(out, self.norm.weight, self.norm.bias, self.norm.eps)
        return F.relu(out.permute(0, 1, 4, 2, 3))


class Predictor(nn.Module):
    """ref/models/Predictor.py:265-359."""

    def __init__(self, max_H, max_W, max_T, h_list, w_list, to_list, tp_list, embed_dim=512,
                 fuse_method='SPADE', param_free_norm_type='layer', evt_hidden_channels=256, evt_n_layers=1,
                 stochastic=True, transformer_layers=4, num_heads=8, window_size=4, dropout=0.1, drop_path=0.1,
                 Spatial_FFN_hidden_ratio=4, dim_feedforward=1024, norm=None, return_intermediate=False,
                 evt_former=True, learn_evt_token=False, evt_former_num_layers=4, rand_context=False):
        super().__init__()
        if norm is None:     # the reference default is ONE shared nn.LayerNorm(512) instance
            norm = nn.LayerNorm(512)
        assert evt_former and not learn_evt_token, "the learned event token branch is out of scope"
        self.stochastic, self.evt_former = stochastic, evt_former
        self.h_list, self.w_list = h_list, w_list
        self.coor_generator = CoorGenerator(max_H, max_W, max_T)
        if not rand_context:
            self.register_buffer("observed_coor", self.coor_generator(to_list, h_list, w_list))
            self.register_buffer("predict_coor", self.coor_generator(tp_list, h_list, w_list))
        else:       # ref :281-284: coordinates are chosen per batch from all_coor
            self.observed_coor = self.predict_coor = None
            self.register_buffer("all_coor", self.coor_generator(torch.cat([to_list, tp_list]), h_list, w_list)
                                 .reshape(max_T, max_H, max_W, 3))
        self.nrmlp = NRMLP(out_channels=embed_dim, fuse_method=fuse_method)
        self.fuser = PosFeatFuser(x_channels=embed_dim, param_free_norm_type=param_free_norm_type)
        self.EVT_Former = VidHRFormerEncoder(evt_former_num_layers, max_H, max_W, embed_dim, num_heads, window_size,
                                             dropout, drop_path, Spatial_FFN_hidden_ratio, dim_feedforward, norm,
                                             learn_evt_token)
        self.evt_posterior = EventEncoder(embed_dim, evt_hidden_channels, evt_n_layers, stochastic)
        self.evt_prior = EventEncoder(embed_dim, evt_hidden_channels, evt_n_layers, stochastic) if stochastic else None
        self.TP = tp_list.shape[0]
        self.transformer = VidHRformerDecoderNAR(transformer_layers, max_H, max_W, embed_dim, num_heads, window_size,
                                                 dropout, drop_path, Spatial_FFN_hidden_ratio, dim_feedforward, norm,
                                                 return_intermediate)

    def _pos(self, coor):
        beta, gamma = self.nrmlp(coor)
        return (beta, gamma if self.nrmlp.fuse_method == 'SPADE' else None)

    def forward(self, observed_features, predict_features_gt=None):
        op, pp = self._pos(self.observed_coor), self._pos(self.predict_coor)
        observed_features, obs_evt = self.evt_coding_forward(observed_features, *op)
        if self.stochastic:
            zo, mu_o, logvar_o = self.evt_prior(obs_evt)
            if predict_features_gt is not None:
                _, pred_evt = self.evt_coding_forward(predict_features_gt, *pp)
                zp, mu_p, logvar_p = self.evt_posterior(pred_evt)
            if self.training:
                assert predict_features_gt is not None, \
                    "please input groundtruth predict features for storchastic model training/val"
                z = zp
            else:
                z = zo
            query_evt = z.unsqueeze(1).repeat(1, self.TP, 1, 1, 1)
            out = self.transformer(query_evt, observed_features, op, pp, self.fuser)
            if predict_features_gt is None:
                return out
            return out, mu_o, logvar_o, mu_p, logvar_p
        mu_o = self.evt_posterior(obs_evt)
        query_evt = mu_o.unsqueeze(1).repeat(1, self.TP, 1, 1, 1)
        return self.transformer(query_evt, observed_features, op, pp, self.fuser)

    def evt_coding_forward(self, x, pos_beta, pos_gamma):
        x = self.EVT_Former(x, (pos_beta, pos_gamma), self.fuser)
        return x, x.mean(dim=1)

    def reset_pos_coor(self, to_list, tp_list):
        dev = self.observed_coor.device if self.observed_coor is not None else self.all_coor.device
        self.predict_coor = self.coor_generator(tp_list, self.h_list, self.w_list).to(dev)
        self.observed_coor = self.coor_generator(to_list, self.h_list, self.w_list).to(dev)
        self.TP = tp_list.shape[0]


class L1Loss(nn.Module):
    """ref/models/criterion.py:99-121 (norm_dim=None path)."""

    def __init__(self, norm_dim=None, lam=1.0):
        super().__init__()
        assert norm_dim is None
        self.lam = lam

    def __call__(self, gt, pred):
        return torch.abs(pred - gt).mean() * self.lam


class Div_KL(nn.Module):
    """ref/models/criterion.py:341-354."""

    def __init__(self, beta):
        super().__init__()
        self.beta = beta

    def forward(self, mu1, logvar1, mu2, logvar2):
        N = mu1.shape[0]
        sigma1, sigma2 = torch.exp(0.5 * logvar1), torch.exp(0.5 * logvar2)
        kld = torch.log(sigma2 / sigma1) + (torch.exp(logvar1) + (mu1 - mu2) ** 2) / (2 * torch.exp(logvar2)) - 0.5
        return self.beta * kld.sum() / N


def build_predictor_from_cfg(cls, P, num_past, num_future, **overrides):
    """Construct `cls` (oracle or HIP Predictor) the way LitPredictor.__init__ does
    (ref/models/Predictor.py:28-47) from the `Predictor:` section of a YAML config."""
    h = torch.linspace(0, P['max_H'] - 1, P['max_H'])
    w = torch.linspace(0, P['max_W'] - 1, P['max_W'])
    to = torch.linspace(0, num_past - 1, num_past)
    tp = torch.linspace(num_past, num_past + num_future - 1, num_future)
    assert P['max_T'] == num_past + num_future, "Incompatible max_T and clip length"
    return cls(P['max_H'], P['max_W'], P['max_T'], h, w, to, tp, P['embed_dim'], P['fuse_method'],
               P['param_free_norm_type'], P['evt_hidden_channels'], 1, P['stochastic'], P['transformer_layers'],
               evt_former=P['evt_former'], learn_evt_token=False,
               evt_former_num_layers=P['evt_former_num_layers'], rand_context=P['rand_context'], **overrides)


def predictor_train_step(predictor, opt, past_feats, future_feats, lam_PF_L1=0.01, KL_beta=1e-8,
                         max_grad_norm=1.0, frozen_dec=None, future_frames=None):
    """One optimisation step = LitPredictor.training_step_no_gan + shared_step
    (ref/models/Predictor.py:124-148,172-194) restated without Lightning, taking
    the frozen encoder's features as input.  With `frozen_dec` given the image L1
    term is included (full-step flavour); otherwise only the feature-L1 + KL
    terms (predictor-only flavour).  Returns a dict of scalars."""
    predictor.zero_grad()
    if predictor.stochastic:
        pred, mu_o, lv_o, mu_p, lv_p = predictor(past_feats, future_feats)
        kl = Div_KL(KL_beta)(mu_o, lv_o, mu_p, lv_p)
    else:
        pred = predictor(past_feats)
        kl = torch.zeros((), dtype=pred.dtype)
    pf = L1Loss(lam=lam_PF_L1)(pred, future_feats)
    loss = pf + kl
    img = None
    if frozen_dec is not None:
        img = L1Loss()(frozen_dec(pred), future_frames)
        loss = loss + img
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(predictor.transformer.parameters(), max_norm=max_grad_norm, norm_type=2)
    opt.step()
    return {"loss": float(loss.detach()), "PF_L1": float(pf.detach()), "KL": float(kl.detach()), "grad_norm": float(gn),
            "Image_L1": None if img is None else float(img.detach())}


def full_train_step(predictor, opt, enc, dec, past_frames, future_frames, lam_PF_L1=0.01, KL_beta=1e-8, max_grad_norm=1.0):
    """Complete Stage-2 step from pixels (ref/models/Predictor.py:124-148,172-194): frozen encoder (no_grad, eval) on
    past and future frames -> predictor -> frozen decoder (eval; gradient flows through it) -> L1(img) + lam*L1(feat) + KL.
    `enc` / `dec` are the frozen autoencoder modules (stock torch; the caller passes them in)."""
    enc.eval(); dec.eval()
    with torch.no_grad():
        past_feats, future_feats = enc(past_frames), enc(future_frames)
    return predictor_train_step(predictor, opt, past_feats, future_feats, lam_PF_L1, KL_beta, max_grad_norm,
                                frozen_dec=dec, future_frames=future_frames)


def predictor_val_step(predictor, past_feats, future_feats, lam_PF_L1=0.01, KL_beta=1e-8, frozen_dec=None, future_frames=None):
    """LitPredictor.validation_step + shared_step (ref/models/Predictor.py:150-170,172-194) restated without Lightning,
    taking the frozen encoder's features: Lightning runs it with the module in eval mode under no_grad, so a stochastic
    predictor is still handed the ground-truth target features (shared_step :181-183), runs BOTH encoders, returns the
    5-tuple and decodes from the PRIOR sample zo (ref :312-321); dropout / drop-path are off and the EventEncoder's
    BatchNorm uses its running statistics.  Scalars as the reference logs them (*_val), plus the prediction."""
    was_training = predictor.training
    predictor.eval()
    try:
        with torch.no_grad():
            if predictor.stochastic:
                pred, mu_o, lv_o, mu_p, lv_p = predictor(past_feats, future_feats)
                kl = Div_KL(KL_beta)(mu_o, lv_o, mu_p, lv_p)
            else:
                pred = predictor(past_feats)
                kl = torch.zeros((), dtype=pred.dtype)
            pf = L1Loss(lam=lam_PF_L1)(pred, future_feats)
            loss = pf + kl
            img = None
            if frozen_dec is not None:
                img = L1Loss()(frozen_dec(pred), future_frames)
                loss = loss + img
    finally:
        predictor.train(was_training)
    return {"loss": float(loss), "PF_L1": float(pf), "KL": float(kl), "Image_L1": None if img is None else float(img),
            "pred": pred}


def full_val_step(predictor, enc, dec, past_frames, future_frames, lam_PF_L1=0.01, KL_beta=1e-8):
    """validation_step from pixels (ref/models/Predictor.py:150-170,172-194)."""
    enc.eval(); dec.eval()
    with torch.no_grad():
        past_feats, future_feats = enc(past_frames), enc(future_frames)
    return predictor_val_step(predictor, past_feats, future_feats, lam_PF_L1, KL_beta, frozen_dec=dec, future_frames=future_frames)


def rand_context_batch_process(predictor, batch):
    """ref/models/Predictor.py:241-251: point observed_coor / predict_coor / TP at this batch's time-steps."""
    clip_o, clip_p, idx_o, idx_p = batch
    coor = predictor.all_coor
    predictor.observed_coor = coor[idx_o].flatten(0, 2)
    predictor.predict_coor = coor[idx_p].flatten(0, 2)
    predictor.TP = idx_p.shape[0]
    return clip_o, clip_p
