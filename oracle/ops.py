"""Functional CPU restatement (plain torch ops) of every operator on the NPVP
Stage-2 predictor hot path.  TEST INFRASTRUCTURE - see oracle/__init__.py.

Canonical activation layout used by the whole build (oracle and HIP path):

    x[F, P, C]   F = N*T frames (f = n*T + t), P = H*W pixels (p = h*W + w),
                 C = embed_dim channels, C contiguous.

The reference instead ping-pongs (N,T,C,H,W) <-> (N,T,H,W,C) <-> (T,N*H*W,C)
<-> (ws*ws, N*T*nwin, C) <-> NCHW (ref/models/VidHRFormer.py:34,50,94,114,
283-307,379,392; ref/models/submodules.py:444-454).  Every function cites the
reference lines whose arithmetic it restates.
"""
import math

import torch
import torch.nn.functional as F

__all__ = [
    "layernorm", "posfuse", "linear", "gelu", "spatial_groups",
    "temporal_groups", "attn_core", "encoder_temporal_mask", "frame_ln",
    "dwconv3x3", "to_canonical", "from_canonical", "key_hashed_fill",
    "synth_features", "seeded_randn", "golden_view",
]


def to_canonical(x):
    """(N,T,C,H,W) -> (N*T, H*W, C).  ref/models/VidHRFormer.py:34,137-138."""
    N, T, C, H, W = x.shape
    return x.permute(0, 1, 3, 4, 2).reshape(N * T, H * W, C).contiguous()


def from_canonical(x, N, T, H, W):
    """(N*T, H*W, C) -> (N,T,C,H,W).  ref/models/VidHRFormer.py:50,159."""
    C = x.shape[-1]
    return x.reshape(N, T, H, W, C).permute(0, 1, 4, 2, 3).contiguous()


def layernorm(x, w, b, eps=1e-5):
    """nn.LayerNorm(C) over the channel dim (ref/models/VidHRFormer.py:65-77)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def posfuse(x, T, beta, gamma=None, add=None, norm="layer", eps=1e-5):
    """PosFeatFuser.forward (ref/models/submodules.py:432-454).

    x: [N*T, P, C]; beta/gamma: [T*P, C] (row order t,h,w - CoorGenerator,
    ref/models/submodules.py:357-364); add: optional [N, P, C] that is added to
    every time-step of sample n BEFORE the normalisation (the `+ query_evt`
    of ref/models/VidHRFormer.py:211,236, query_evt being z repeated over T,
    ref/models/Predictor.py:317,332).
    'layer' = GroupNorm(1, C, affine=False) on (N*T, C, H, W): statistics over
    all C*H*W elements of one frame, biased variance, eps 1e-5
    (ref/models/submodules.py:427,446).
    """
    Fr, P, C = x.shape
    N = Fr // T
    u = x
    if add is not None:
        u = (x.reshape(N, T, P, C) + add.reshape(N, 1, P, C)).reshape(Fr, P, C)
    if norm == "layer":
        mu = u.mean(dim=(1, 2), keepdim=True)
        var = ((u - mu) ** 2).mean(dim=(1, 2), keepdim=True)
    elif norm == "instance":   # InstanceNorm2d(affine=False): per (frame, channel)
        mu = u.mean(dim=1, keepdim=True)
        var = ((u - mu) ** 2).mean(dim=1, keepdim=True)
    else:
        raise ValueError(f"{norm} is not a supported param-free norm type")
    xh = (u - mu) * torch.rsqrt(var + eps)
    xh = xh.reshape(N, T, P, C)
    b = beta.reshape(1, T, P, C)
    if gamma is not None:
        xh = xh * (1 + gamma.reshape(1, T, P, C))
    return (xh + b).reshape(Fr, P, C)


def linear(x, w, b=None):
    return F.linear(x, w, b)


def gelu(x):
    """nn.GELU() default = exact erf form (ref/models/VidHRFormer.py:73,337)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def spatial_groups(Fr, H, W, ws):
    """Row indices of every non-overlapping ws x ws window.

    Returns LongTensor [Fr * (H/ws) * (W/ws), ws*ws]: group (f, qh, qw), member
    (ph, pw) -> canonical row f*H*W + (qh*ws+ph)*W + qw*ws+pw.  This is the
    rearrange "n (qh ph) (qw pw) c -> (ph pw) (n qh qw) c" of
    ref/models/VidHRFormer.py:453-462 expressed as index math.  H, W must be
    multiples of ws (true for every config: 8 / 4; the centre-pad branch
    ref/models/VidHRFormer.py:488-500 is never taken).
    """
    assert H % ws == 0 and W % ws == 0, "window must tile the feature grid"
    f = torch.arange(Fr).view(Fr, 1, 1, 1, 1)
    qh = torch.arange(H // ws).view(1, -1, 1, 1, 1)
    qw = torch.arange(W // ws).view(1, 1, -1, 1, 1)
    ph = torch.arange(ws).view(1, 1, 1, -1, 1)
    pw = torch.arange(ws).view(1, 1, 1, 1, -1)
    rows = f * (H * W) + (qh * ws + ph) * W + (qw * ws + pw)
    return rows.reshape(-1, ws * ws)


def temporal_groups(N, T, P):
    """Row indices of every per-pixel time strip: [N*P, T], group (n, p), member
    t -> canonical row (n*T + t)*P + p.  Restates the
    `permute(1,0,2,3,4).reshape(T, N*H*W, C)` of ref/models/VidHRFormer.py:94,217."""
    n = torch.arange(N).view(N, 1, 1)
    p = torch.arange(P).view(1, P, 1)
    t = torch.arange(T).view(1, 1, T)
    return ((n * T + t) * P + p).reshape(N * P, T)


def encoder_temporal_mask(T):
    """Bool [T, T], True = NOT allowed: every query except the last may not see
    the last time-step (ref/models/VidHRFormer.py:100-102)."""
    m = torch.zeros(T, T, dtype=torch.bool)
    m[0:-1, -1] = True
    return m


def attn_core(q, k, v, q_rows, k_rows, num_heads, mask=None, p_drop=0.0,
              training=False):
    """softmax(q k^T / sqrt(d) + mask) v per group and head.

    q: [Rq, C] (already projected, NOT yet scaled), k, v: [Rk, C];
    q_rows [G, L], k_rows [G, S] index the rows of each attention group.
    Restates torch.nn.MultiheadAttention's slow path as called at
    ref/models/VidHRFormer.py:104-107,221,239,298-300: q scaled by d**-0.5,
    bool mask -> -inf, softmax over keys, dropout on the weights (train only),
    heads concatenated.  Returns o [Rq, C] at the query rows.
    """
    C = q.shape[-1]
    d = C // num_heads
    G, L = q_rows.shape
    S = k_rows.shape[1]
    qg = q[q_rows.reshape(-1)].reshape(G, L, num_heads, d).permute(0, 2, 1, 3)
    kg = k[k_rows.reshape(-1)].reshape(G, S, num_heads, d).permute(0, 2, 1, 3)
    vg = v[k_rows.reshape(-1)].reshape(G, S, num_heads, d).permute(0, 2, 1, 3)
    s = (qg * (d ** -0.5)) @ kg.transpose(-1, -2)
    if mask is not None:
        s = s.masked_fill(mask.view(1, 1, L, S), float("-inf"))
    p = torch.softmax(s, dim=-1)
    if training and p_drop > 0.0:
        p = F.dropout(p, p_drop, True)
    og = (p @ vg).permute(0, 2, 1, 3).reshape(G * L, C)
    o = torch.empty(q.shape[0], C, dtype=q.dtype)
    o[q_rows.reshape(-1)] = og
    return o


def frame_ln(h, w, b, eps=1e-5):
    """nn.LayerNorm((Ch, H, W)) of MlpDWBN (ref/models/VidHRFormer.py:348,361,367):
    statistics over all Ch*H*W elements of one frame, per-element affine.
    h: [F, P, Ch]; w, b: [P, Ch] (the state-dict tensors are (Ch, H, W); the
    caller transposes them)."""
    mu = h.mean(dim=(1, 2), keepdim=True)
    var = ((h - mu) ** 2).mean(dim=(1, 2), keepdim=True)
    return (h - mu) * torch.rsqrt(var + eps) * w + b


def dwconv3x3(h, w, b, H, W):
    """Depthwise 3x3, stride 1, zero pad 1 (ref/models/VidHRFormer.py:351-358).
    h: [F, P, Ch]; w: [Ch, 3, 3]; b: [Ch]."""
    Fr, P, Ch = h.shape
    x = h.reshape(Fr, H, W, Ch).permute(0, 3, 1, 2)
    y = F.conv2d(x, w.reshape(Ch, 1, 3, 3), b, stride=1, padding=1, groups=Ch)
    return y.permute(0, 2, 3, 1).reshape(Fr, P, Ch)


def golden_view(t, limit=65536, stride=5):
    """How a tensor is stored in / compared with tests/golden/*.npz: whole when it has
    at most `limit` elements, else every `stride`-th element of the flattened tensor."""
    t = t.detach()
    return t if t.numel() <= limit else t.flatten()[::stride]


def seeded_randn(shape, seed):
    """randn from a private CPU generator (never touches the global RNG)."""
    g = torch.Generator().manual_seed(int(seed) & 0x7FFFFFFF)
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32)


def synth_features(shape, seed):
    """Synthetic post-ReLU encoder features relu(N(0.05, 0.1^2)) (the
    real frozen ResnetEncoder emits >=0 features, ref/models/ResNetAutoEncoder.py:118,142)."""
    return torch.relu(seeded_randn(shape, seed) * 0.1 + 0.05)


def key_hashed_fill(module, seed=0):
    """Deterministic, construction-order-independent weight fill shared by the
    reference (golden generation), the oracle and the HIP modules.
    For each state_dict key: Generator(crc32(key) ^ seed) -> randn.  Norm / BN
    scales are 1 + 0.1 randn, biases 0.02 randn, nrmlp.B 10 randn, >=2-D
    weights fan-in scaled, running_var positive.  `EVT_Former.norm.*` is the
    same tensor as `transformer.norm.*` (ref/models/Predictor.py:270,290,299):
    it is skipped so that `transformer.norm.*` wins.
    """
    import zlib
    sd = module.state_dict()
    with torch.no_grad():
        for key, t in sd.items():
            if key.endswith("num_batches_tracked") or key.endswith("_coor"):
                continue
            if key.startswith("EVT_Former.norm.") and "transformer.norm.weight" in sd:
                continue
            g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
            r = torch.randn(t.shape, generator=g, dtype=torch.float32)
            leaf = key.split(".")[-1]
            parent = key.split(".")[-2] if "." in key else ""
            # 1-D weights are LayerNorm/BatchNorm scales; the (Ch,H,W) frame-LN
            # scales of MlpDWBN live under a parent called norm1/2/3.
            is_norm = t.dim() == 1 or parent.startswith("norm")
            if leaf == "running_var":
                val = 0.5 + r.abs()
            elif leaf == "running_mean":
                val = 0.1 * r
            elif key.endswith("nrmlp.B") or key == "B":
                val = 10.0 * r
            elif leaf == "gamma":          # NonLocalAttenion2D skip gain of the frozen autoencoder (reference init: 0)
                val = 0.1 * r
            elif leaf == "bias" or leaf == "in_proj_bias":
                val = 0.02 * r
            elif is_norm and leaf == "weight":
                val = 1.0 + 0.1 * r
            elif t.dim() >= 2:
                fan_in = t[0].numel()
                val = r / math.sqrt(fan_in)
            else:
                val = 1.0 + 0.1 * r
            t.copy_(val)
    return module
