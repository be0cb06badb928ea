"""Re-create the inputs of every golden fixture (tests/golden/make_golden.py) from
their seeds and run an implementation on them.  `impl` is a namespace exposing the
reference's class names (the oracle package on CPU, or npvp_amd on cuda:0), so the
same case code checks the oracle against the reference's vectors and the HIP path
against them too.
"""
import os

import numpy as np
import torch

from oracle import ops as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double().cpu().flatten()
    b = torch.as_tensor(b).detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_row_rel_err(a, b, floor=1e-6):
    """Worst relative L2 error of a ROW (a vector along the last axis: a token row of an activation or of its gradient, an
    output-feature row of a weight gradient) - the whole-tensor rel-L2 is blind to a few small rows that are wrong.  Rows whose
    reference norm is below `floor` x the largest row norm are not counted (where rows mix - attention, frame norms, residual
    sums - such a row is below the fp32 resolution of the tensor it was computed from, in the reference's own arithmetic too);
    floor = 0 counts every non-zero row (pure GEMMs: a row of the product depends on that row of the operand only).
    Returns 0 for tensors with fewer than two axes or a last axis shorter than 32 (an (N,T,C,8,8) feature map's last axis is 8
    pixels of one channel, a 3 x 3 kernel's is 3 taps: a norm over so few - often post-ReLU zero - values says nothing)."""
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    if b.dim() < 2 or a.shape != b.shape or b.shape[-1] < 32:
        return 0.0
    a, b = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
    nb = b.norm(dim=1)
    keep = nb > max(floor * float(nb.max()), 0.0)
    keep &= nb > 0
    if not bool(keep.any()):
        return 0.0
    return float(((a - b).norm(dim=1)[keep] / nb[keep]).max())


ERR_LOG = os.environ.get("NPVP_ERR_LOG")      # optional: append "<tag> <key> <rel-L2> <worst row>" lines (GPU triage)


def log_err(tag, key, e, er=None):
    if ERR_LOG:
        with open(ERR_LOG, "a") as f:
            f.write(f"{tag} {key} {e:.3e}" + (f" row {er:.3e}" if er is not None else "") + "\n")


def compare(results, golden, tol, skip=(), tag="", row_tol=None):
    """Every golden array (except meta) must be reproduced within rel-L2 `tol`, and - for arrays stored with their row
    structure - every ROW of it within `row_tol` (max_row_rel_err).  north_star's bar is 1e-3; the default here is 2e-5 wherever
    the whole-tensor bound is the usual 1e-4 (measured on the GPU, every mode: <= 7.6e-6, the exact fp32-MFMA mode included - that
    much is the oracle's own fp32 rounding against the reference), 10 x tol for cases with a looser whole-tensor bound."""
    if row_tol is None:
        row_tol = 2e-5 if tol <= 1e-4 else 10 * tol
    worst = 0.0
    errs, rerrs = {}, {}
    for key, g in golden.items():
        if key == "meta" or key in skip:
            continue
        assert key in results, f"case does not produce golden key {key}"
        mine = O.golden_view(results[key]) if results[key].numel() != g.size else results[key]
        e = rel_err(mine, g)
        er = max_row_rel_err(mine, g) if tuple(torch.as_tensor(mine).shape) == tuple(g.shape) else 0.0
        worst = max(worst, e)
        errs[key] = e
        rerrs[key] = er
        log_err(tag, key, e, er)
    bad = {k: f"{v:.3e}" for k, v in errs.items() if not v < tol}
    assert not bad, f"rel-L2 above {tol:.1e}: {bad}"
    bad = {k: f"{v:.3e}" for k, v in rerrs.items() if not v < row_tol}
    assert not bad, f"worst-row rel-L2 above {row_tol:.1e}: {bad}"
    return worst


def pos_tables(T, seed, with_gamma, dev):
    beta = (0.5 * O.seeded_randn((T * 64, 512), seed)).to(dev)
    gamma = (0.3 * O.seeded_randn((T * 64, 512), seed + 1)).to(dev) if with_gamma else None
    return beta, gamma


def _g(out, cot, inputs):
    return torch.autograd.grad((out * cot).sum(), inputs)


def case_posfuse(impl, dev, norm="layer"):
    N, T = 2, 3
    x = O.seeded_randn((N, T, 8, 8, 512), 11).to(dev).requires_grad_()
    beta, gamma = pos_tables(T, 12, True, dev)
    beta.requires_grad_(); gamma.requires_grad_()
    cot = O.seeded_randn((N, T, 8, 8, 512), 14).to(dev)
    fz = impl.PosFeatFuser(512, norm)
    y = fz(x, beta, gamma)
    gx, gb, gg = _g(y, cot, [x, beta, gamma])
    add = O.seeded_randn((N, 8, 8, 512), 15).to(dev).requires_grad_()
    y2 = fz(x, beta, gamma, add=add)
    gx2, ga2 = _g(y2, cot, [x, add])
    return dict(y=y, gx=gx, gbeta=gb, ggamma=gg, y_add=y2, gx_add=gx2, gadd=ga2)


def case_nrmlp(impl, dev, fuse="Add"):
    m = impl.NRMLP(512, fuse_method=fuse)
    O.key_hashed_fill(m, int(load(f"nrmlp_{fuse}")["meta"][0]))   # seed chosen by make_golden (ReLU-kink margin)
    m = m.to(dev)
    coor = impl.CoorGenerator(8, 8, 7)(torch.linspace(3, 6, 4), torch.linspace(0, 7, 8), torch.linspace(0, 7, 8)).to(dev)
    b, g = m(coor)
    cot = O.seeded_randn(b.shape, 22).to(dev)
    gB = torch.autograd.grad((b * cot).sum() + (g * cot).sum(), m.B)[0]
    return dict(coor=coor, beta=b, gamma=g, gB=gB)


def case_slmhsa(impl, dev):
    N, T = 1, 2
    m = impl.SpatialLocalMultiheadAttention(512, 8, 4, 0.0)
    O.key_hashed_fill(m, 31)
    m = m.to(dev)
    x = O.seeded_randn((N, T, 8, 8, 512), 32).to(dev).requires_grad_()
    v = O.seeded_randn((N, T, 8, 8, 512), 33).to(dev).requires_grad_()
    cot = O.seeded_randn((N, T, 8, 8, 512), 34).to(dev)
    y = m(x, value=v)
    gx, gv, gw = _g(y, cot, [x, v, m.attn.in_proj_weight])
    return dict(y=y, gx=gx, gv=gv, gW_rows=gw[::64])


def _fg(norm_module, g):
    """gradient of a frame-LayerNorm parameter in the reference's (Ch,H,W) layout (the HIP modules store that
    parameter channels-last and expose `ref_view`)"""
    return norm_module.ref_view(g) if hasattr(norm_module, "ref_view") else g


def case_mlpdwbn(impl, dev):
    N, T = 1, 2
    m = impl.MlpDWBN(8, 8, 512, 2048, 512, drop=0.0)
    O.key_hashed_fill(m, 41)
    m = m.to(dev)
    x = O.seeded_randn((N, T, 8, 8, 512), 42).to(dev).requires_grad_()
    cot = O.seeded_randn((N, T, 8, 8, 512), 43).to(dev)
    y = m(x)
    ps = [m.norm1.weight, m.dw3x3.weight, m.norm3.bias, m.fc2.weight, m.fc1.bias, m.dw3x3.bias, m.norm2.weight]
    g = _g(y, cot, [x] + ps)
    return dict(y=y, gx=g[0], g_norm1_w=_fg(m.norm1, g[1])[::16], g_dw_w=g[2], g_norm3_b=_fg(m.norm3, g[3])[::8],
                g_fc2_w_rows=g[4].flatten(1)[::32], g_fc1_b=g[5], g_dw_b=g[6], g_norm2_w=_fg(m.norm2, g[7])[::16])


def case_block_enc(impl, dev):
    N, T = 1, 3
    m = impl.VidHRFormerBlockEnc(8, 8, 512, 8, 4, 0.0, 0.0, 4, 1024)
    O.key_hashed_fill(m, 51)
    m = m.to(dev)
    x = O.synth_features((N, T, 8, 8, 512), 52).to(dev).requires_grad_()
    beta, _ = pos_tables(T, 53, False, dev)
    cot = O.seeded_randn((N, T, 8, 8, 512), 54).to(dev)
    y = m(x, (beta, None), impl.PosFeatFuser(512, 'layer'))
    gx, gn3, gl1 = _g(y, cot, [x, m.norm3.weight, m.linear1.weight])
    return dict(y=y, gx=gx, g_norm3_w=gn3, g_linear1_w_rows=gl1[::64])


def case_block_dec(impl, dev):
    N, T2, T1 = 1, 3, 2
    m = impl.VidHRFormerBlockDecNAR(8, 8, 512, 8, 4, 0.0, 0.0, 4, 1024)
    O.key_hashed_fill(m, 61)
    m = m.to(dev)
    tgt = (0.3 * O.seeded_randn((N, T2, 8, 8, 512), 62)).to(dev).requires_grad_()
    qe = (0.5 * O.seeded_randn((N, 8, 8, 512), 63)).to(dev).requires_grad_()
    mem = O.synth_features((N, T1, 8, 8, 512), 64).to(dev).requires_grad_()
    mb, _ = pos_tables(T1, 65, False, dev)
    tb, _ = pos_tables(T2, 66, False, dev)
    cot = O.seeded_randn((N, T2, 8, 8, 512), 67).to(dev)
    y = m(tgt, qe, mem, (mb, None), (tb, None), impl.PosFeatFuser(512, 'layer'))
    g = _g(y, cot, [tgt, qe, mem, m.EncDecAttn.in_proj_weight, m.norm5.bias])
    return dict(y=y, gtgt=g[0], gqe=g[1], gmem=g[2], g_encdec_W_rows=g[3][::64], g_norm5_b=g[4])


def case_evtenc(impl, dev):
    N = 3
    m = impl.EventEncoder(512, 256, 1, True)
    O.key_hashed_fill(m, 71)
    m = m.to(dev)
    x = O.synth_features((N, 512, 8, 8), 72).to(dev)
    eps = O.seeded_randn((N, 512, 8, 8), 73).to(dev)
    m.eps_fn = lambda shape: eps
    out = {}
    for mode in ("train", "eval"):
        m.train(mode == "train")
        with torch.no_grad():
            z, mu, lv = m(x)
        out.update({f"z_{mode}": z, f"mu_{mode}": mu, f"logvar_{mode}": lv})
    out["running_mean_conv1"] = m.conv1[1].running_mean
    return out


def case_losses(impl, dev):
    a = O.seeded_randn((2, 4, 512, 8, 8), 81).to(dev); b = O.seeded_randn((2, 4, 512, 8, 8), 82).to(dev)
    mu1, lv1 = O.seeded_randn((2, 512, 8, 8), 83).to(dev), (0.3 * O.seeded_randn((2, 512, 8, 8), 84)).to(dev)
    mu2, lv2 = O.seeded_randn((2, 512, 8, 8), 85).to(dev), (0.3 * O.seeded_randn((2, 512, 8, 8), 86)).to(dev)
    return dict(l1=impl.L1Loss(lam=0.01)(a, b), kl=impl.Div_KL(1e-6)(mu1, lv1, mu2, lv2))


def _small_predictor(impl, stochastic, seed, dev, evt_layers=2, dec_layers=2, To=3, Tp=4, **kw):
    h = torch.linspace(0, 7, 8)
    to, tp = torch.linspace(0, To - 1, To), torch.linspace(To, To + Tp - 1, Tp)
    args = dict(evt_former=True, learn_evt_token=False, evt_former_num_layers=evt_layers, rand_context=False,
                dropout=0.0, drop_path=0.0)
    args.update(kw)
    m = impl.Predictor(8, 8, To + Tp, h, h, to, tp, 512, 'Add', 'layer', 256, 1, stochastic, dec_layers, **args)
    O.key_hashed_fill(m, seed)
    return m.to(dev)


def case_predictor(impl, dev, variant="D"):
    stochastic = variant == "S"
    N, To, Tp = 2, 3, 4
    m = _small_predictor(impl, stochastic, 91, dev)
    past = O.synth_features((N, To, 512, 8, 8), 92).to(dev)
    fut = O.synth_features((N, Tp, 512, 8, 8), 93).to(dev)
    eps = O.seeded_randn((N, 512, 8, 8), 94).to(dev)
    cot = O.seeded_randn((N, Tp, 512, 8, 8), 95).to(dev)
    if stochastic:
        m.evt_prior.eps_fn = m.evt_posterior.eps_fn = lambda shape: eps
    res = {}
    m.eval()
    with torch.no_grad():
        res["y_eval"] = m(past)
    m.train()
    p = past.clone().requires_grad_()
    o = m(p, fut) if stochastic else m(p)
    yt = o[0] if stochastic else o
    loss = (yt * yt * cot).sum()          # smooth at the final ReLU's kink (see make_golden.py)
    if stochastic:
        loss = loss + impl.Div_KL(1e-2)(*o[1:])
        for i, n_ in enumerate(["mu_o", "logvar_o", "mu_p", "logvar_p"]):
            res[n_] = o[1 + i]
    m.zero_grad()
    loss.backward()
    sd = dict(m.named_parameters())
    res.update(y_train=yt, g_past=p.grad, g_tied_norm_w=m.transformer.norm.weight.grad,
               g_tied_norm_b=m.transformer.norm.bias.grad, g_nrmlp_B=m.nrmlp.B.grad,
               g_sffn1_norm2_w=_fg(m.transformer.layers[1].SpatialFFN1.norm2,
                                   sd["transformer.layers.1.SpatialFFN1.norm2.weight"].grad)[::16],
               g_evt_tmhsa_W_rows=sd["EVT_Former.layers.0.temporal_MHSA.in_proj_weight"].grad[::64],
               g_encdec_out_W_rows=sd["transformer.layers.0.EncDecAttn.out_proj.weight"].grad[::32],
               g_post_conv2_w=sd["evt_posterior.conv2.0.weight"].grad[::8, ::8])
    res["decoder_grad_norm"] = torch.sqrt(sum((q.grad ** 2).sum() for q in m.transformer.parameters()))
    return res


def case_train_step(impl, dev, variant="D", make_opt=None):
    """Two steps of the predictor-only training step (ref Predictor.py:124-148,172-194)."""
    stochastic = variant == "S"
    N, To, Tp = 2, 3, 4
    m = _small_predictor(impl, stochastic, 101, dev)
    past = O.synth_features((N, To, 512, 8, 8), 92).to(dev)
    fut = O.synth_features((N, Tp, 512, 8, 8), 93).to(dev)
    eps = O.seeded_randn((N, 512, 8, 8), 94).to(dev)
    if stochastic:
        m.evt_prior.eps_fn = m.evt_posterior.eps_fn = lambda shape: eps
    m.train()
    opt = make_opt(m) if make_opt is not None else torch.optim.AdamW(m.parameters(), lr=1e-4)
    res = {}
    for it in range(2):
        s = impl.predictor_train_step(m, opt, past, fut, 0.01, 1e-6, 1.0)
        res[f"loss_{it}"] = torch.tensor(s["loss"]); res[f"pf_{it}"] = torch.tensor(s["PF_L1"])
        res[f"kl_{it}"] = torch.tensor(s["KL"]); res[f"grad_norm_{it}"] = torch.tensor(s["grad_norm"])
        sd = m.state_dict()
        for n_, key in (("w_dec_lin1", "transformer.layers.1.linear1.weight"), ("w_tied", "transformer.norm.weight"),
                        ("w_evt_fc1", "EVT_Former.layers.0.SpatialFFN.fc1.bias"), ("w_B", "nrmlp.B")):
            res[f"{n_}_{it}"] = sd[key].flatten()[:256].clone()
    return res


def case_predictor_full(impl, dev):
    N, To, Tp = 1, 2, 3
    m = _small_predictor(impl, False, 111, dev, evt_layers=4, dec_layers=8, To=To, Tp=Tp, dropout=0.1, drop_path=0.1)
    m.eval()
    past = O.synth_features((N, To, 512, 8, 8), 112).to(dev)
    with torch.no_grad():
        y = m(past)
    return dict(y_strided=y.flatten()[::7], y_mean=y.mean(), y_std=y.std())


def case_ae(impl, dev, tag="64", device_layout=None):
    """Frozen autoencoder: encoder features, decoder frames, decoder input-gradient.  device_layout = the product's
    to_device_layout (channels_last encoder + BatchNorm folding + fused epilogue kernels)."""
    ci, ngf, nd, nres, S, out_layer = {"64": (1, 64, 3, 2, 64, 'Sigmoid'), "128": (3, 32, 4, 3, 128, 'Tanh')}[tag]
    enc = impl.ResnetEncoder(ci, ngf=ngf, n_downsampling=nd, num_res_blocks=nres, learn_3d=False)
    dec = impl.ResnetDecoder(ci, ngf=ngf, n_downsampling=nd, out_layer=out_layer)
    O.key_hashed_fill(enc, 121); O.key_hashed_fill(dec, 122)
    enc, dec = enc.eval(), dec.eval()
    if device_layout is None:
        enc, dec = enc.to(dev), dec.to(dev)
    else:
        for q in list(enc.parameters()) + list(dec.parameters()):
            q.requires_grad_(False)
        enc, dec = device_layout(enc, dec, dev)
    x = torch.rand(1, 2, ci, S, S, generator=torch.Generator().manual_seed(123)).to(dev)
    with torch.no_grad():
        feats = enc(x)
    f = feats.clone().requires_grad_()
    y = dec(f)
    cot = O.seeded_randn(y.shape, 124).to(dev)
    (y * cot).sum().backward()
    return dict(feats=feats, frames=y, g_feats=f.grad)


def case_full_step(pred_impl, ae_impl, dev, make_opt=None, device_layout=None):
    """One complete Stage-2 step from pixels: frozen enc -> predictor -> frozen dec -> L1(img) + 0.01 L1(feat) + KL."""
    N, To, Tp = 2, 3, 4
    m = _small_predictor(pred_impl, True, 131, dev)
    enc = ae_impl.ResnetEncoder(1, ngf=64, n_downsampling=3, num_res_blocks=2, learn_3d=False)
    dec = ae_impl.ResnetDecoder(1, ngf=64, n_downsampling=3, out_layer='Sigmoid')
    O.key_hashed_fill(enc, 121); O.key_hashed_fill(dec, 122)
    enc, dec = (enc.to(dev), dec.to(dev)) if device_layout is None else device_layout(enc, dec, dev)
    enc, dec = enc.eval(), dec.eval()
    for q in list(enc.parameters()) + list(dec.parameters()):
        q.requires_grad_(False)
    g_ = torch.Generator().manual_seed(133)
    pf, ff = torch.rand(N, To, 1, 64, 64, generator=g_).to(dev), torch.rand(N, Tp, 1, 64, 64, generator=g_).to(dev)
    eps = O.seeded_randn((N, 512, 8, 8), 134).to(dev)
    m.evt_prior.eps_fn = m.evt_posterior.eps_fn = lambda shape: eps
    m.train()
    opt = make_opt(m) if make_opt is not None else torch.optim.AdamW(m.parameters(), lr=1e-4)
    s = pred_impl.full_train_step(m, opt, enc, dec, pf, ff, 0.01, 1e-6, 1.0)
    sd = m.state_dict()
    return dict(loss=torch.tensor(s["loss"]), img=torch.tensor(s["Image_L1"]), pf=torch.tensor(s["PF_L1"]),
                kl=torch.tensor(s["KL"]), grad_norm=torch.tensor(s["grad_norm"]),
                w_dec_lin1=sd["transformer.layers.1.linear1.weight"].flatten()[:256].clone(),
                w_evt_fc1=sd["EVT_Former.layers.0.SpatialFFN.fc1.bias"].flatten()[:256].clone())


def case_val_step(pred_impl, ae_impl, dev, variant="S", device_layout=None):
    """validation_step (ref Predictor.py:150-170): the step is entered with the module in TRAIN mode and must itself run in
    eval mode under no_grad (NPVP-S: with the ground truth, 5-tuple, decoded from the prior sample) and restore the mode."""
    stochastic = variant == "S"
    N, To, Tp = 2, 3, 4
    m = _small_predictor(pred_impl, stochastic, 151, dev, dropout=0.1, drop_path=0.1)
    enc = ae_impl.ResnetEncoder(1, ngf=64, n_downsampling=3, num_res_blocks=2, learn_3d=False)
    dec = ae_impl.ResnetDecoder(1, ngf=64, n_downsampling=3, out_layer='Sigmoid')
    O.key_hashed_fill(enc, 121); O.key_hashed_fill(dec, 122)
    for q in list(enc.parameters()) + list(dec.parameters()):
        q.requires_grad_(False)
    enc, dec = (enc.to(dev), dec.to(dev)) if device_layout is None else device_layout(enc, dec, dev)
    enc, dec = enc.eval(), dec.eval()
    g_ = torch.Generator().manual_seed(153)
    pf, ff = torch.rand(N, To, 1, 64, 64, generator=g_).to(dev), torch.rand(N, Tp, 1, 64, 64, generator=g_).to(dev)
    eps = O.seeded_randn((N, 512, 8, 8), 154).to(dev)
    if stochastic:
        m.evt_prior.eps_fn = m.evt_posterior.eps_fn = lambda shape: eps
    m.train()
    s = pred_impl.full_val_step(m, enc, dec, pf, ff, 0.01, 1e-6)
    assert m.training, "the validation step must restore the module's training mode"
    with torch.no_grad():
        past_feats, fut_feats = enc(pf), enc(ff)
        frames = dec(s["pred"])
    sp = pred_impl.predictor_val_step(m, past_feats, fut_feats, 0.01, 1e-6)
    return dict(loss=torch.tensor(s["loss"]), img=torch.tensor(s["Image_L1"]), pf=torch.tensor(s["PF_L1"]), kl=torch.tensor(s["KL"]),
                pred=s["pred"], frames_strided=frames.flatten()[::5], loss_features_only=torch.tensor(sp["loss"]))


def case_randctx(impl, dev):
    """Unified model: Predictor(rand_context=True) with an unsorted random context / target split of T=7 steps."""
    N, T = 2, 7
    h = torch.linspace(0, 7, 8)
    tl = torch.linspace(0, T - 1, T)
    m = impl.Predictor(8, 8, T, h, h, tl[:3], tl[3:], 512, 'Add', 'layer', 256, 1, True, 2, evt_former=True,
                       learn_evt_token=False, evt_former_num_layers=2, rand_context=True, dropout=0.0, drop_path=0.0)
    O.key_hashed_fill(m, int(load("predictor_randctx_S")["meta"][2]))   # seed chosen by make_golden (ReLU-kink margin)
    m = m.to(dev)
    clip = O.synth_features((N, T, 512, 8, 8), 142).to(dev)
    idx_o, idx_p = torch.tensor([4, 0, 6, 2]), torch.tensor([5, 1, 3])
    xo, xp = impl.rand_context_batch_process(m, (clip[:, idx_o.to(dev)], clip[:, idx_p.to(dev)], idx_o.to(dev), idx_p.to(dev)))
    eps = O.seeded_randn((N, 512, 8, 8), 143).to(dev)
    cot = O.seeded_randn((N, 3, 512, 8, 8), 144).to(dev)
    m.evt_prior.eps_fn = m.evt_posterior.eps_fn = lambda shape: eps
    m.train()
    x = xo.clone().requires_grad_()
    y = m(x, xp)
    (y[0] * y[0] * cot).sum().backward()
    return dict(out=y[0], mu_o=y[1], mu_p=y[3], g_obs=x.grad, gB=m.nrmlp.B.grad, g_tied=m.transformer.norm.weight.grad)


def case_fractime(impl, dev):
    """Continuous time: a deterministic predictor built for integer steps, re-pointed with reset_pos_coor at fractional
    context / target times (and a different number of targets)."""
    N, T = 2, 7
    h = torch.linspace(0, 7, 8)
    tl = torch.linspace(0, T - 1, T)
    m = impl.Predictor(8, 8, T, h, h, tl[:3], tl[3:], 512, 'SPADE', 'layer', 256, 1, False, 2, evt_former=True,
                       learn_evt_token=False, evt_former_num_layers=2, dropout=0.0, drop_path=0.0)
    O.key_hashed_fill(m, 151)
    m = m.to(dev)
    m.reset_pos_coor(torch.tensor([0.0, 1.5, 3.25]), torch.tensor([3.75, 4.5, 5.0, 6.5, 7.0]))
    feats = O.synth_features((N, 3, 512, 8, 8), 152).to(dev)
    cot = O.seeded_randn((N, 5, 512, 8, 8), 153).to(dev)
    m.train()
    x = feats.clone().requires_grad_()
    y = m(x)
    (y * y * cot).sum().backward()
    return dict(out=y, g_obs=x.grad, gB=m.nrmlp.B.grad, coor_p=m.predict_coor)
