"""Frozen Stage-1 autoencoder around the predictor (ref/models/ResNetAutoEncoder.py:51-261 and the
`Factorized3DConvAttn` / `NonLocalAttenion2D` blocks of ref/models/submodules.py:9-180), needed for the
FULL training step of Stage 2: frozen encoder on past+future frames (no grad), frozen decoder on the
predicted features with the gradient flowing through it to the predictor (ref/models/Predictor.py:172-194).

SURVEY 8f "next" row #1.  Same class names, constructor signatures and state-dict keys as the reference, so Stage-1
checkpoints load.  Only the `learn_3d=False` configuration (every shipped config) is supported.
  stage 1: the modules as built run on stock PyTorch-ROCm (MIOpen convolutions, rocBLAS matmuls);
  stage 2: `to_device_layout` -> `fuse_frozen_autoencoder`: the pair is frozen and in eval mode in Stage 2, so every
           conv -> BatchNorm -> ReLU (-> + skip) group becomes a `FoldedConvAct`: the convolution with the BatchNorm folded
           into its weights (MIOpen) + ONE hand-written epilogue pass `npvp_bias_act` (csrc/ae.hip) instead of separate
           bias / BatchNorm / ReLU / skip-add passes; the decoder's input gradient goes through `npvp_act_bwd`.
"""
import functools

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


_ACT_CODE = {nn.ReLU: 1, nn.Tanh: 2, nn.Sigmoid: 3}


def _bn_scale_shift(bn):
    s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return s, bn.bias - bn.running_mean * s


class FoldedConvAct(nn.Module):
    """[pad ->] conv / transposed conv [-> BatchNorm2d (eval)] [-> ReLU / Tanh / Sigmoid] of a FROZEN network as one convolution
    with folded weights (MIOpen, no bias) and one epilogue pass out = act(conv + bias[c]) (+ residual)  (csrc/ae.hip)."""

    def __init__(self, conv, bn=None, act=None, pad=None):
        super().__init__()
        assert isinstance(conv, (nn.Conv2d, nn.ConvTranspose2d)) and conv.groups == 1
        self.transposed = isinstance(conv, nn.ConvTranspose2d)
        w = conv.weight.detach()
        b = conv.bias.detach() if conv.bias is not None else torch.zeros(conv.out_channels, dtype=w.dtype, device=w.device)
        if bn is not None:
            assert isinstance(bn, nn.BatchNorm2d) and not bn.training and bn.track_running_stats, "fold needs an eval-mode BatchNorm2d"
            s, t = _bn_scale_shift(bn)
            w = w * (s.view(1, -1, 1, 1) if self.transposed else s.view(-1, 1, 1, 1))
            b = b * s + t
        fmt = torch.channels_last if conv.weight.is_contiguous(memory_format=torch.channels_last) and not conv.weight.is_contiguous() \
            else torch.contiguous_format
        self.register_buffer("weight", w.contiguous(memory_format=fmt))
        self.register_buffer("bias", b.detach().float().contiguous())
        self.stride, self.padding, self.dilation = conv.stride, conv.padding, conv.dilation
        self.output_padding = conv.output_padding if self.transposed else None
        self.act = 0 if act is None else _ACT_CODE[type(act)]
        self.pad = pad

    def forward(self, x, residual=None):
        if self.pad is not None:
            x = self.pad(x)
        if self.transposed:
            y = F.conv_transpose2d(x, self.weight, None, self.stride, self.padding, self.output_padding, 1, self.dilation)
        else:
            y = F.conv2d(x, self.weight, None, self.stride, self.padding, self.dilation, 1)
        return ops.bias_act(y, self.bias, self.act, residual)


def _fold_sequential(seq):
    """Sequential of [pad] conv [BatchNorm2d] [activation] groups -> Sequential of FoldedConvAct"""
    out, mods, i = [], list(seq), 0
    while i < len(mods):
        pad = None
        if isinstance(mods[i], (nn.ReflectionPad2d, nn.ReplicationPad2d)):
            pad = mods[i]; i += 1
        conv = mods[i]; i += 1
        assert isinstance(conv, (nn.Conv2d, nn.ConvTranspose2d)), f"cannot fold {type(conv).__name__}"
        bn = act = None
        if i < len(mods) and isinstance(mods[i], nn.BatchNorm2d):
            bn = mods[i]; i += 1
        if i < len(mods) and type(mods[i]) in _ACT_CODE:
            act = mods[i]; i += 1
        out.append(FoldedConvAct(conv, bn, act, pad))
    return nn.Sequential(*out)


def fuse_frozen_autoencoder(enc, dec):
    """In-place module surgery on an eval-mode pair that already holds its final weights; freezes its parameters (do it after load_state_dict:
    the folded modules no longer carry the reference's state-dict keys).  Returns (enc, dec)."""
    for m in (enc, dec):
        assert not m.training, "fuse_frozen_autoencoder: eval mode only (BatchNorm must use its running statistics)"
        for q in m.parameters():
            q.requires_grad_(False)          # Stage 2 never trains the pair (ref/models/Predictor.py:17-25)
    with torch.no_grad():
        for name, mod in list(enc.named_children()):
            if isinstance(mod, nn.Sequential):
                setattr(enc, name, _fold_sequential(mod))
            elif isinstance(mod, Factorized3DConvAttn):
                mod.spatial_conv = _fold_sequential(mod.spatial_conv)[0]
                a = mod.attn2d
                if isinstance(a.norm_func, nn.BatchNorm2d):      # fold the BatchNorm behind out_proj into the Linear
                    s, t = _bn_scale_shift(a.norm_func)
                    a.out_proj.weight.mul_(s.view(-1, 1))
                    a.out_proj.bias.copy_(a.out_proj.bias * s + t)
                    a.norm_func = nn.Identity()
            elif isinstance(mod, ResnetBlock):
                mod.conv_block = _fold_sequential(mod.conv_block)
        dec.model = _fold_sequential(dec.model)
    return enc, dec


class NonLocalAttenion2D(nn.Module):
    """Self-attention over the H*W positions of one frame with 2x2 max-pooled keys/values
    (ref/models/submodules.py:98-176).  (sic: the reference spells it 'Attenion'.)"""

    def __init__(self, in_channels, atten_channels_downsample_ratio=8, value_channels_downsample_ratio=2, bias=True,
                 learn_gamma=True, norm_func=None, activ_func=None):
        super().__init__()
        self.bias, self.in_channels = bias, in_channels
        self.attn_dim = in_channels // atten_channels_downsample_ratio
        self.value_dim = in_channels // value_channels_downsample_ratio
        self.Wq = nn.Linear(in_channels, self.attn_dim, bias=bias)
        self.Wk = nn.Linear(in_channels, self.attn_dim, bias=bias)
        self.Wv = nn.Linear(in_channels, self.value_dim, bias=bias)
        self.out_proj = nn.Linear(self.value_dim, in_channels, bias=bias)
        self.max_pool = nn.MaxPool2d((2, 2), stride=2)
        self.learn_gamma = learn_gamma
        self.gamma = nn.Parameter(torch.tensor(0., dtype=torch.float32)) if learn_gamma else 1.0
        self.norm_func = norm_func if norm_func is not None else nn.Identity()
        self.activ_func = activ_func if activ_func is not None else nn.Identity()
        for lin in (self.Wq, self.Wk, self.Wv, self.out_proj):
            nn.init.xavier_uniform_(lin.weight)
            if bias:
                nn.init.constant_(lin.bias, 0.)

    def forward(self, x):
        N, C, H, W = x.shape
        tok = x.flatten(2).transpose(1, 2)                                        # (N, HW, C)
        q = self.Wq(tok)                                                          # (N, HW, a)
        k = self.max_pool(self.Wk(tok).transpose(1, 2).reshape(N, self.attn_dim, H, W)).flatten(2)      # (N, a, HW/4)
        v = self.max_pool(self.Wv(tok).transpose(1, 2).reshape(N, self.value_dim, H, W)).flatten(2)     # (N, v, HW/4)
        att = F.softmax(q @ k, dim=-1)                                            # un-scaled scores, as the reference
        out = self.out_proj(att @ v.transpose(1, 2))                              # (N, HW, C)
        out = out.transpose(1, 2).reshape(N, C, H, W)
        return x + self.gamma * self.activ_func(self.norm_func(out))


class Factorized3DConvAttn(nn.Module):
    """ref/models/submodules.py:9-95 with learn_3d=False: conv3x3+BN+ReLU (+skip) -> NonLocalAttenion2D -> +skip."""

    def __init__(self, in_channels, atten_channels_downsample_ratio=8, value_channels_downsample_ratio=2, use_bias=True,
                 learn_gamma=True, norm_layer_2d=nn.BatchNorm2d, norm_layer_1d=nn.BatchNorm1d, activ_func=nn.ReLU(),
                 conv_first=True, learn_3d=True):
        super().__init__()
        if learn_3d:
            raise NotImplementedError("learn_3d=True (temporal conv + 1-D attention) is not used by any shipped config")
        self.in_channels, self.learn_3d, self.conv_first = in_channels, learn_3d, conv_first
        self.spatial_conv = nn.Sequential(nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1, bias=use_bias),
                                          norm_layer_2d(in_channels), activ_func)
        self.attn2d = NonLocalAttenion2D(in_channels, atten_channels_downsample_ratio, value_channels_downsample_ratio, True,
                                         learn_gamma, norm_layer_2d(in_channels), activ_func=activ_func)
        self.temporal_conv = None
        self.attn1d = None

    def forward(self, x, T):
        if isinstance(self.spatial_conv, FoldedConvAct):       # fused: relu(conv + b) + x in the epilogue pass
            if self.conv_first:
                return self.attn2d(self.spatial_conv(x, residual=x)) + x
            y = self.attn2d(x)
            return self.spatial_conv(y, residual=y) + x
        if self.conv_first:
            return self.attn2d(self.spatial_conv(x) + x) + x
        y = self.attn2d(x)
        return self.spatial_conv(y) + y + x


class ResnetBlock(nn.Module):
    """ref/models/ResNetAutoEncoder.py:206-261: x + [pad, conv3, norm, ReLU, (dropout), pad, conv3, norm](x)."""

    def __init__(self, dim, padding_type, norm_layer, use_dropout, use_bias):
        super().__init__()
        pads = {'reflect': nn.ReflectionPad2d, 'replicate': nn.ReplicationPad2d}
        layers = []
        for half in range(2):
            p = 0
            if padding_type in pads:
                layers.append(pads[padding_type](1))
            elif padding_type == 'zero':
                p = 1
            else:
                raise NotImplementedError('padding [%s] is not implemented' % padding_type)
            layers += [nn.Conv2d(dim, dim, kernel_size=3, padding=p, bias=use_bias), norm_layer(dim)]
            if half == 0:
                layers.append(nn.ReLU(True))
                if use_dropout:
                    layers.append(nn.Dropout(0.5))
        self.conv_block = nn.Sequential(*layers)

    def forward(self, x):
        if len(self.conv_block) == 2 and isinstance(self.conv_block[1], FoldedConvAct):      # fused: skip-add in the epilogue
            return self.conv_block[1](self.conv_block[0](x), residual=x)
        return x + self.conv_block(x)


def _use_bias(norm_layer):
    if type(norm_layer) == functools.partial:
        return norm_layer.func == nn.InstanceNorm2d
    return norm_layer == nn.InstanceNorm2d


class ResnetEncoder(nn.Module):
    """ref/models/ResNetAutoEncoder.py:51-146: (N,T,Ci,S,S) frames -> (N,T,ngf*2^n_down,S/2^n_down,S/2^n_down), ReLU'd."""

    def __init__(self, input_nc, ngf=64, n_downsampling=3, num_res_blocks=2, norm_layer=nn.BatchNorm2d,
                 norm_layer1d=nn.BatchNorm1d, use_dropout=False, padding_type='reflect', learn_3d=True):
        super().__init__()
        use_bias = _use_bias(norm_layer)
        self.n_downsampling, self.num_res_blocks = n_downsampling, num_res_blocks
        self.block0 = nn.Sequential(nn.ReflectionPad2d(3), nn.Conv2d(input_nc, ngf, kernel_size=7, padding=0, bias=use_bias),
                                    norm_layer(ngf), nn.ReLU(True))
        self.block1 = nn.Sequential(nn.Conv2d(ngf, ngf * 2, kernel_size=3, stride=2, padding=1, bias=use_bias),
                                    norm_layer(ngf * 2), nn.ReLU(True))
        ch = ngf * 2
        mk_attn = lambda c: Factorized3DConvAttn(in_channels=c, norm_layer_2d=norm_layer, norm_layer_1d=norm_layer1d,
                                                 activ_func=nn.ReLU(True), learn_3d=learn_3d)
        for i in range(1, n_downsampling):
            setattr(self, f'block{i + 1}_3dConvAttn', mk_attn(ch))
            setattr(self, f'block{i + 1}_conv', nn.Sequential(
                nn.Conv2d(ch, ch * 2, kernel_size=3, stride=2, padding=1, bias=use_bias), norm_layer(ch * 2), nn.ReLU(True)))
            ch *= 2
        for i in range(num_res_blocks):
            setattr(self, f'res_3dConvAttn_{i}', mk_attn(ch))
            setattr(self, f'res_conv_{i}', ResnetBlock(ch, padding_type=padding_type, norm_layer=norm_layer,
                                                       use_dropout=use_dropout, use_bias=use_bias))
        self.out_act = nn.ReLU()

    def forward(self, x):
        N, T = x.shape[:2]
        x = self.block1(self.block0(x.flatten(0, 1)))
        for i in range(1, self.n_downsampling):
            x = getattr(self, f'block{i + 1}_conv')(getattr(self, f'block{i + 1}_3dConvAttn')(x, T))
        for i in range(self.num_res_blocks):
            x = getattr(self, f'res_conv_{i}')(getattr(self, f'res_3dConvAttn_{i}')(x, T))
        x = self.out_act(x)
        return x.reshape(N, T, *x.shape[1:])


class ResnetDecoder(nn.Module):
    """ref/models/ResNetAutoEncoder.py:148-204: n_downsampling x [ConvTranspose2d s2 + norm + ReLU], 7x7 conv, Tanh/Sigmoid."""

    def __init__(self, output_nc, ngf=64, n_downsampling=2, norm_layer=nn.BatchNorm2d, use_dropout=False,
                 padding_type='reflect', out_layer='Tanh'):
        super().__init__()
        use_bias = _use_bias(norm_layer)
        model = []
        for i in range(n_downsampling):
            mult = 2 ** (n_downsampling - i)
            model += [nn.ConvTranspose2d(ngf * mult, ngf * mult // 2, kernel_size=3, stride=2, padding=1, output_padding=1,
                                         bias=use_bias), norm_layer(ngf * mult // 2), nn.ReLU(True)]
        model += [nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_nc, kernel_size=7, padding=0)]
        if out_layer == 'Tanh':
            model.append(nn.Tanh())
        elif out_layer == 'Sigmoid':
            model.append(nn.Sigmoid())
        else:
            raise ValueError("Unsupported output layer")
        self.model = nn.Sequential(*model)

    def forward(self, x):
        N, T = x.shape[:2]
        y = self.model(x.flatten(0, 1))
        return y.reshape(N, T, *y.shape[1:])


def build_frozen_autoencoder(AE, img_channels):
    """(encoder, decoder) as LitPredictor.__init__ prepares them (ref/models/Predictor.py:17-25): built from the `AE:`
    section of a reference YAML, parameters frozen, eval mode (BatchNorm uses running statistics)."""
    enc = ResnetEncoder(img_channels, ngf=AE['ngf'], n_downsampling=AE['n_downsampling'], num_res_blocks=AE['num_res_blocks'],
                        norm_layer=nn.BatchNorm2d, norm_layer1d=nn.BatchNorm1d, learn_3d=AE['learn_3d'])
    dec = ResnetDecoder(img_channels, ngf=AE['ngf'], n_downsampling=AE['n_downsampling'], out_layer=AE['out_layer'],
                        norm_layer=nn.BatchNorm2d)
    for m in (enc, dec):
        for p in m.parameters():
            p.requires_grad_(False)
        m.eval()
    return enc, dec


def to_device_layout(enc, dec, device):
    """Move the frozen pair to an MI355X.  The ENCODER runs in torch.channels_last (MIOpen's NHWC kernels: 27.5 -> 24.8 ms
    for 640 frames of 64x64, and its (N*T,H,W,C) output is the predictor's canonical layout, so the layout transpose at
    the predictor's entry disappears); the decoder stays NCHW (NHWC measured slower: 9.0 -> 11.2 ms fwd + input-grad)."""
    enc, dec = enc.to(device).to(memory_format=torch.channels_last), dec.to(device)
    return fuse_frozen_autoencoder(enc.eval(), dec.eval())
