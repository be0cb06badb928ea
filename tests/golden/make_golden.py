#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the REFERENCE
(`/root/reference/models`, CPU, torch 2.10) and, in the same run, report how far
the oracle restatement (oracle/) is from it.

Dev-container only: /root/reference does not exist on the GPU box, so nothing in
tests/, smoke() or bench.py runs this script - they read the committed .npz
files.  Only DATA is written (inputs are re-generated from seeds; outputs,
gradients and scalars are stored); no reference source is copied.

The reference needs two absent third-party modules which are stubbed exactly as
SURVEY 8c describes: timm.models.layers.to_2tuple and
pytorch_lightning.LightningModule.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from oracle import ops as O  # noqa: E402


def import_reference():
    timm = types.ModuleType('timm')
    tm = types.ModuleType('timm.models')
    tl = types.ModuleType('timm.models.layers')
    tl.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    sys.modules.update({'timm': timm, 'timm.models': tm, 'timm.models.layers': tl})
    pl = types.ModuleType('pytorch_lightning')

    class LightningModule(nn.Module):
        pass
    pl.LightningModule = LightningModule
    sys.modules['pytorch_lightning'] = pl
    sys.path.insert(0, '/root/reference')
    import models as ref_models
    import models.VidHRFormer as ref_vid
    import models.submodules as ref_sub
    return ref_models, ref_vid, ref_sub


def import_reference_metrics():
    """ref/utils/metrics.py executed as-is.  The `utils` package's __init__ pulls in the dataset module (torchvision, cv2:
    absent here), so the file is loaded as a member of an empty stand-in package whose `train_summary` sibling supplies the
    one name it imports (`load_ckpt`, only used when a checkpoint path is passed)."""
    import importlib.util
    pkg = types.ModuleType('refutils')
    pkg.__path__ = ['/root/reference/utils']
    ts = types.ModuleType('refutils.train_summary')
    ts.load_ckpt = None
    sys.modules.update({'refutils': pkg, 'refutils.train_summary': ts})
    spec = importlib.util.spec_from_file_location('refutils.metrics', '/root/reference/utils/metrics.py')
    mod = importlib.util.module_from_spec(spec)
    sys.modules['refutils.metrics'] = mod
    spec.loader.exec_module(mod)
    return mod


REPORT = []


def rel_err(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check(name, mine, ref, tol=1e-5):
    e = rel_err(mine, ref)
    REPORT.append((name, e))
    print(f"  {name:58s} oracle-vs-reference rel-L2 = {e:.3e}")
    assert e < tol, f"oracle deviates from the reference on {name}: {e}"


def npy(t):
    return O.golden_view(t).cpu().numpy().astype(np.float32)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def pos_tables(T, seed, with_gamma):
    beta = 0.5 * O.seeded_randn((T * 64, 512), seed)
    gamma = 0.3 * O.seeded_randn((T * 64, 512), seed + 1) if with_gamma else torch.zeros(T * 64, 512)
    return beta, gamma


def grads(out, cot, inputs):
    gs = torch.autograd.grad((out * cot).sum(), inputs, allow_unused=False)
    return gs


def gen_val_steps(R):
    """LitPredictor.validation_step (ref/models/Predictor.py:150-170) = shared_step (:172-194) in eval mode under no_grad
    (what Lightning's validation loop does) + the three losses, restated on the reference's own modules: NPVP-S is handed the
    ground-truth target features, returns the 5-tuple and decodes from the prior sample zo (:312-321)."""
    import npvp_amd
    real_randn = torch.randn
    h = torch.linspace(0, 7, 8)
    N, To, Tp = 2, 3, 4
    to, tp = torch.linspace(0, To - 1, To), torch.linspace(To, To + Tp - 1, Tp)
    kw = dict(evt_former=True, learn_evt_token=False, evt_former_num_layers=2, rand_context=False)     # dropout / drop_path at their 0.1 defaults: eval must ignore them
    renc = R.ResnetEncoder(1, ngf=64, n_downsampling=3, num_res_blocks=2, learn_3d=False)
    rdec = R.ResnetDecoder(1, ngf=64, n_downsampling=3, out_layer='Sigmoid')
    menc = npvp_amd.ResnetEncoder(1, ngf=64, n_downsampling=3, num_res_blocks=2, learn_3d=False)
    mdec = npvp_amd.ResnetDecoder(1, ngf=64, n_downsampling=3, out_layer='Sigmoid')
    for m_, sd in ((renc, 121), (menc, 121), (rdec, 122), (mdec, 122)):
        O.key_hashed_fill(m_, sd); m_.eval()
        for p_ in m_.parameters():
            p_.requires_grad_(False)
    g_ = torch.Generator().manual_seed(153)
    pf, ff = torch.rand(N, To, 1, 64, 64, generator=g_), torch.rand(N, Tp, 1, 64, 64, generator=g_)
    eps = O.seeded_randn((N, 512, 8, 8), 154)
    for variant, stochastic in (("D", False), ("S", True)):
        ref = R.Predictor(8, 8, To + Tp, h, h, to, tp, 512, 'Add', 'layer', 256, 1, stochastic, 2, norm=nn.LayerNorm(512), **kw)
        mine = oracle.Predictor(8, 8, To + Tp, h, h, to, tp, 512, 'Add', 'layer', 256, 1, stochastic, 2, **kw)
        O.key_hashed_fill(ref, 151); O.key_hashed_fill(mine, 151)
        if stochastic:
            mine.evt_prior.eps_fn = mine.evt_posterior.eps_fn = lambda shape: eps
        # reference side: Lightning's validation loop = module.eval() + no_grad around validation_step
        ref.eval()
        with torch.no_grad():
            past_feats, fut_feats = renc(pf), renc(ff)
            torch.randn = lambda *a, **k: eps
            try:
                o = ref(past_feats, fut_feats) if stochastic else ref(past_feats)
            finally:
                torch.randn = real_randn
            pred = o[0] if stochastic else o
            kl = R.Div_KL(1e-6)(*o[1:]) if stochastic else torch.zeros(())
            frames = rdec(pred)
            img = R.L1Loss()(frames, ff); pfl = R.L1Loss(lam=0.01)(pred, fut_feats)
            loss = img + pfl + kl
        mine.train()                          # the step itself must switch to eval (and back)
        sm = oracle.full_val_step(mine, menc, mdec, pf, ff, 0.01, 1e-6)
        assert mine.training
        check(f"val_step[{variant}].loss", torch.tensor(sm["loss"]), loss); check(f"val_step[{variant}].Image_L1", torch.tensor(sm["Image_L1"]), img)
        check(f"val_step[{variant}].PF_L1", torch.tensor(sm["PF_L1"]), pfl); check(f"val_step[{variant}].pred", sm["pred"], pred)
        if stochastic:
            check(f"val_step[{variant}].KL", torch.tensor(sm["KL"]), kl)
        # predictor-only flavour on the same features (what the bench's predictor-only step validates with)
        sp = oracle.predictor_val_step(mine, past_feats, fut_feats, 0.01, 1e-6)
        check(f"val_step[{variant}].features_only.loss", torch.tensor(sp["loss"]), pfl + kl)
        save(f"val_step_{variant}", loss=npy(loss), img=npy(img), pf=npy(pfl), kl=npy(kl), pred=npy(pred), frames_strided=npy(frames.flatten()[::5]),
             loss_features_only=npy(pfl + kl), meta=np.array([N, To, Tp, 151, 121, 122, 153, 154]))


def write_report(append=False):
    path = os.path.join(HERE, "ORACLE_VS_REFERENCE.txt")
    if append:
        keep = [l for l in open(path).read().splitlines() if l and not l.startswith("val_step")]
        with open(path, "w") as f:
            f.write("\n".join(keep) + "\n")
            for n_, e in REPORT:
                f.write(f"{n_:64s} {e:.3e}\n")
    else:
        with open(path, "w") as f:
            f.write("# oracle (CPU restatement) vs imported reference, rel-L2, torch %s, generated by make_golden.py\n" % torch.__version__)
            for n_, e in REPORT:
                f.write(f"{n_:64s} {e:.3e}\n")
    print(f"max oracle-vs-reference rel-L2 over {len(REPORT)} checks: {max(e for _, e in REPORT):.3e}")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R, RV, RS = import_reference()
    if "val" in sys.argv[1:]:                 # only the validation-step fixtures (the others regenerate bit-identically anyway)
        gen_val_steps(R)
        write_report(append=True)
        return

    # ------------------------------------------------------------------ posfuse
    for norm in ("layer", "instance"):
        N, T = 2, 3
        x = O.seeded_randn((N, T, 8, 8, 512), 11).requires_grad_()
        beta, gamma = pos_tables(T, 12, True)
        beta.requires_grad_(); gamma.requires_grad_()
        cot = O.seeded_randn((N, T, 8, 8, 512), 14)
        ref = RS.PosFeatFuser(512, norm)
        y = ref(x, beta, gamma)
        gx, gb, gg = grads(y, cot, [x, beta, gamma])
        mine = oracle.PosFeatFuser(512, norm)
        ym = mine(x, beta, gamma)
        gxm, gbm, ggm = grads(ym, cot, [x, beta, gamma])
        for n_, a, b in (("y", ym, y), ("gx", gxm, gx), ("gbeta", gbm, gb), ("ggamma", ggm, gg)):
            check(f"posfuse[{norm}].{n_}", a, b)
        # decoder form: fuse(x + query_evt)
        add = O.seeded_randn((N, 8, 8, 512), 15).requires_grad_()
        y2 = ref(x + add.unsqueeze(1), beta, gamma)
        gx2, ga2 = grads(y2, cot, [x, add])
        y2m = mine(x, beta, gamma, add=add)
        gx2m, ga2m = grads(y2m, cot, [x, add])
        check(f"posfuse[{norm}]+add.y", y2m, y2); check(f"posfuse[{norm}]+add.gadd", ga2m, ga2)
        save(f"posfuse_{norm}", y=npy(y), gx=npy(gx), gbeta=npy(gb), ggamma=npy(gg),
             y_add=npy(y2), gx_add=npy(gx2), gadd=npy(ga2), meta=np.array([N, T, 11, 12, 14, 15]))

    # ------------------------------------------------------------------ NRMLP o CoorGenerator
    # The MLP has ReLUs: a hidden unit whose pre-activation is within rounding noise of 0 makes d/dB
    # discontinuous (one flipped unit moves gB by ~1e-3).  Pick the first fill seed with no unit closer than
    # 1e-5 to the kink (fp32-grade paths differ by ~5e-7 there), so every arithmetic path (fp32 / split-precision MFMA) sits on the same side.
    def nrmlp_margin(mod, coor):
        x = mod.gaussian_mapping(coor)
        worst = 1e9
        for l in mod.MLP:
            x = l(x)
            if isinstance(l, nn.Linear):
                worst = min(worst, float(x.abs().min()))
        return worst
    for fuse in ("Add", "SPADE"):
        cg = R.CoorGenerator(8, 8, 7)
        coor = cg(torch.linspace(3, 6, 4), torch.linspace(0, 7, 8), torch.linspace(0, 7, 8))
        for seed in range(21, 40000, 100):
            ref = R.NRMLP(512, fuse_method=fuse)
            O.key_hashed_fill(ref, seed)
            with torch.no_grad():
                mg = nrmlp_margin(ref, coor)
            if mg > 1e-5:
                break
        print(f"  nrmlp[{fuse}]: fill seed {seed}, smallest |pre-activation| {mg:.2e}")
        b, g = ref(coor)
        mine = oracle.NRMLP(512, fuse_method=fuse)
        O.key_hashed_fill(mine, seed)
        coor_m = oracle.CoorGenerator(8, 8, 7)(torch.linspace(3, 6, 4), torch.linspace(0, 7, 8), torch.linspace(0, 7, 8))
        bm, gm = mine(coor_m)
        check(f"coor[{fuse}]", coor_m, coor); check(f"nrmlp[{fuse}].beta", bm, b)
        cot = O.seeded_randn(b.shape, 22)
        gB = torch.autograd.grad((b * cot).sum() + (g * cot).sum(), ref.B)[0]
        gBm = torch.autograd.grad((bm * cot).sum() + (gm * cot).sum(), mine.B)[0]
        check(f"nrmlp[{fuse}].gB", gBm, gB)
        save(f"nrmlp_{fuse}", coor=npy(coor), beta=npy(b), gamma=npy(g), gB=npy(gB), meta=np.array([seed, 22]))

    # ------------------------------------------------------------------ spatial window MHA
    N, T = 1, 2
    ref = RV.SpatialLocalMultiheadAttention(512, 8, 4, 0.0)
    O.key_hashed_fill(ref, 31)
    mine = oracle.SpatialLocalMultiheadAttention(512, 8, 4, 0.0)
    O.key_hashed_fill(mine, 31)
    x = O.seeded_randn((N, T, 8, 8, 512), 32).requires_grad_()
    v = O.seeded_randn((N, T, 8, 8, 512), 33).requires_grad_()
    cot = O.seeded_randn((N, T, 8, 8, 512), 34)
    y = ref(x, value=v)
    gx, gv, gw = grads(y, cot, [x, v, ref.attn.in_proj_weight])
    ym = mine(x, value=v)
    gxm, gvm, gwm = grads(ym, cot, [x, v, mine.attn.in_proj_weight])
    for n_, a, b in (("y", ym, y), ("gx", gxm, gx), ("gv", gvm, gv), ("gW", gwm, gw)):
        check(f"slmhsa.{n_}", a, b)
    save("slmhsa", y=npy(y), gx=npy(gx), gv=npy(gv), gW_rows=npy(gw[::64]), meta=np.array([N, T, 31, 32, 33, 34]))

    # ------------------------------------------------------------------ MlpDWBN
    N, T = 1, 2
    ref = RV.MlpDWBN(8, 8, 512, 2048, 512, drop=0.0)
    O.key_hashed_fill(ref, 41)
    mine = oracle.MlpDWBN(8, 8, 512, 2048, 512, drop=0.0)
    O.key_hashed_fill(mine, 41)
    x = O.seeded_randn((N, T, 8, 8, 512), 42).requires_grad_()
    cot = O.seeded_randn((N, T, 8, 8, 512), 43)
    y = ref(x)
    ps = [ref.norm1.weight, ref.dw3x3.weight, ref.norm3.bias, ref.fc2.weight, ref.fc1.bias, ref.dw3x3.bias, ref.norm2.weight]
    pm = [mine.norm1.weight, mine.dw3x3.weight, mine.norm3.bias, mine.fc2.weight, mine.fc1.bias, mine.dw3x3.bias, mine.norm2.weight]
    g = grads(y, cot, [x] + ps)
    ym = mine(x)
    gm = grads(ym, cot, [x] + pm)
    check("mlpdwbn.y", ym, y)
    for n_, a, b in zip(["gx", "g_norm1_w", "g_dw_w", "g_norm3_b", "g_fc2_w", "g_fc1_b", "g_dw_b", "g_norm2_w"], gm, g):
        check(f"mlpdwbn.{n_}", a, b)
    save("mlpdwbn", y=npy(y), gx=npy(g[0]), g_norm1_w=npy(g[1][::16]), g_dw_w=npy(g[2]), g_norm3_b=npy(g[3][::8]),
         g_fc2_w_rows=npy(g[4].flatten(1)[::32]), g_fc1_b=npy(g[5]), g_dw_b=npy(g[6]), g_norm2_w=npy(g[7][::16]),
         meta=np.array([N, T, 41, 42, 43]))

    # ------------------------------------------------------------------ encoder block (mask quirk visible at T>=3)
    N, T = 1, 3
    ref = RV.VidHRFormerBlockEnc(8, 8, 512, 8, 4, 0.0, 0.0, 4, 1024)
    O.key_hashed_fill(ref, 51)
    mine = oracle.VidHRFormerBlockEnc(8, 8, 512, 8, 4, 0.0, 0.0, 4, 1024)
    O.key_hashed_fill(mine, 51)
    x = O.synth_features((N, T, 8, 8, 512), 52).requires_grad_()
    beta, gamma = pos_tables(T, 53, False)
    cot = O.seeded_randn((N, T, 8, 8, 512), 54)
    y = ref(x, (beta, gamma), RS.PosFeatFuser(512, 'layer'))
    gx, gn3, gl1 = grads(y, cot, [x, ref.norm3.weight, ref.linear1.weight])
    ym = mine(x, (beta, None), oracle.PosFeatFuser(512, 'layer'))
    gxm, gn3m, gl1m = grads(ym, cot, [x, mine.norm3.weight, mine.linear1.weight])
    for n_, a, b in (("y", ym, y), ("gx", gxm, gx), ("g_norm3_w", gn3m, gn3), ("g_linear1_w", gl1m, gl1)):
        check(f"block_enc.{n_}", a, b)
    save("block_enc", y=npy(y), gx=npy(gx), g_norm3_w=npy(gn3), g_linear1_w_rows=npy(gl1[::64]),
         meta=np.array([N, T, 51, 52, 53, 54]))

    # ------------------------------------------------------------------ decoder block
    N, T2, T1 = 1, 3, 2
    ref = RV.VidHRFormerBlockDecNAR(8, 8, 512, 8, 4, 0.0, 0.0, 4, 1024)
    O.key_hashed_fill(ref, 61)
    mine = oracle.VidHRFormerBlockDecNAR(8, 8, 512, 8, 4, 0.0, 0.0, 4, 1024)
    O.key_hashed_fill(mine, 61)
    tgt = (0.3 * O.seeded_randn((N, T2, 8, 8, 512), 62)).requires_grad_()
    qe = (0.5 * O.seeded_randn((N, 8, 8, 512), 63)).requires_grad_()
    mem = O.synth_features((N, T1, 8, 8, 512), 64).requires_grad_()
    mb, mg = pos_tables(T1, 65, False)
    tb, tg = pos_tables(T2, 66, False)
    cot = O.seeded_randn((N, T2, 8, 8, 512), 67)
    fz = RS.PosFeatFuser(512, 'layer')
    y = ref(tgt, qe.unsqueeze(1).repeat(1, T2, 1, 1, 1), mem, (mb, mg), (tb, tg), fz)
    g = grads(y, cot, [tgt, qe, mem, ref.EncDecAttn.in_proj_weight, ref.norm5.bias])
    ym = mine(tgt, qe, mem, (mb, None), (tb, None), oracle.PosFeatFuser(512, 'layer'))
    gm = grads(ym, cot, [tgt, qe, mem, mine.EncDecAttn.in_proj_weight, mine.norm5.bias])
    check("block_dec.y", ym, y)
    for n_, a, b in zip(["gtgt", "gqe", "gmem", "g_encdec_W", "g_norm5_b"], gm, g):
        check(f"block_dec.{n_}", a, b)
    save("block_dec", y=npy(y), gtgt=npy(g[0]), gqe=npy(g[1]), gmem=npy(g[2]), g_encdec_W_rows=npy(g[3][::64]),
         g_norm5_b=npy(g[4]), meta=np.array([N, T2, T1, 61, 62, 63, 64, 65, 66, 67]))

    # ------------------------------------------------------------------ EventEncoder (train + eval BN, injected eps)
    N = 3
    ref = RS.EventEncoder(512, 256, 1, True)
    O.key_hashed_fill(ref, 71)
    mine = oracle.EventEncoder(512, 256, 1, True)
    O.key_hashed_fill(mine, 71)
    x = O.synth_features((N, 512, 8, 8), 72)
    eps = O.seeded_randn((N, 512, 8, 8), 73)
    mine.eps_fn = lambda shape: eps
    real_randn = torch.randn
    out = {}
    for mode in ("train", "eval"):
        ref.train(mode == "train"); mine.train(mode == "train")
        torch.randn = lambda *a, **k: eps
        try:
            z, mu, lv = ref(x)
        finally:
            torch.randn = real_randn
        zm, mum, lvm = mine(x)
        check(f"evtenc[{mode}].z", zm, z); check(f"evtenc[{mode}].mu", mum, mu); check(f"evtenc[{mode}].logvar", lvm, lv)
        out.update({f"z_{mode}": npy(z), f"mu_{mode}": npy(mu), f"logvar_{mode}": npy(lv)})
    out["running_mean_conv1"] = npy(ref.conv1[1].running_mean)
    check("evtenc.running_mean", mine.conv1[1].running_mean, ref.conv1[1].running_mean)
    save("evtenc", meta=np.array([N, 71, 72, 73]), **out)

    # ------------------------------------------------------------------ losses
    a = O.seeded_randn((2, 4, 512, 8, 8), 81); b = O.seeded_randn((2, 4, 512, 8, 8), 82)
    mu1, lv1 = O.seeded_randn((2, 512, 8, 8), 83), 0.3 * O.seeded_randn((2, 512, 8, 8), 84)
    mu2, lv2 = O.seeded_randn((2, 512, 8, 8), 85), 0.3 * O.seeded_randn((2, 512, 8, 8), 86)
    l1 = R.L1Loss(lam=0.01)(a, b); kl = R.Div_KL(1e-6)(mu1, lv1, mu2, lv2)
    check("L1Loss", oracle.L1Loss(lam=0.01)(a, b), l1); check("Div_KL", oracle.Div_KL(1e-6)(mu1, lv1, mu2, lv2), kl)
    save("losses", l1=npy(l1), kl=npy(kl), meta=np.array([81, 82, 83, 84, 85, 86]))

    # ------------------------------------------------------------------ whole Predictor, reduced depth, D and S
    h = torch.linspace(0, 7, 8)
    for variant, stochastic in (("D", False), ("S", True)):
        N, To, Tp = 2, 3, 4
        to, tp = torch.linspace(0, To - 1, To), torch.linspace(To, To + Tp - 1, Tp)
        kw = dict(evt_former=True, learn_evt_token=False, evt_former_num_layers=2, rand_context=False,
                  dropout=0.0, drop_path=0.0)
        ref = R.Predictor(8, 8, To + Tp, h, h, to, tp, 512, 'Add', 'layer', 256, 1, stochastic, 2,
                          norm=nn.LayerNorm(512), **kw)
        mine = oracle.Predictor(8, 8, To + Tp, h, h, to, tp, 512, 'Add', 'layer', 256, 1, stochastic, 2, **kw)
        assert list(ref.state_dict().keys()) == list(mine.state_dict().keys()), "state-dict keys differ"
        O.key_hashed_fill(ref, 91); O.key_hashed_fill(mine, 91)
        for (k1, v1), (k2, v2) in zip(ref.state_dict().items(), mine.state_dict().items()):
            assert torch.equal(v1, v2), k1
        past = O.synth_features((N, To, 512, 8, 8), 92)
        fut = O.synth_features((N, Tp, 512, 8, 8), 93)
        eps = O.seeded_randn((N, 512, 8, 8), 94)
        cot = O.seeded_randn((N, Tp, 512, 8, 8), 95)
        if stochastic:
            mine.evt_prior.eps_fn = mine.evt_posterior.eps_fn = lambda shape: eps
        res = {}
        # eval
        ref.eval(); mine.eval()
        torch.randn = lambda *a, **k: eps
        try:
            with torch.no_grad():
                ye = ref(past)
        finally:
            torch.randn = real_randn
        with torch.no_grad():
            yem = mine(past)
        check(f"predictor[{variant}].eval", yem, ye)
        res["y_eval"] = npy(ye)
        # train (dropout=drop_path=0), gradients
        ref.train(); mine.train()
        p1 = past.clone().requires_grad_(); p2 = past.clone().requires_grad_()
        torch.randn = lambda *a, **k: eps
        try:
            o = ref(p1, fut) if stochastic else ref(p1)
        finally:
            torch.randn = real_randn
        om = mine(p2, fut) if stochastic else mine(p2)
        yt, ytm = (o[0], om[0]) if stochastic else (o, om)
        # sum(cot * y^2): elements at the final ReLU's kink (y ~ 0) carry ~0 gradient, so a rounding-level
        # sign difference there cannot move the compared gradients
        loss = (yt * yt * cot).sum(); lossm = (ytm * ytm * cot).sum()
        if stochastic:
            loss = loss + R.Div_KL(1e-2)(*o[1:]); lossm = lossm + oracle.Div_KL(1e-2)(*om[1:])
            for i, n_ in enumerate(["mu_o", "logvar_o", "mu_p", "logvar_p"]):
                check(f"predictor[S].{n_}", om[1 + i], o[1 + i]); res[n_] = npy(o[1 + i])
        ref.zero_grad(); mine.zero_grad()
        loss.backward(); lossm.backward()
        check(f"predictor[{variant}].train", ytm, yt)
        check(f"predictor[{variant}].g_past", p2.grad, p1.grad)
        check(f"predictor[{variant}].g_tied_norm_w", mine.transformer.norm.weight.grad, ref.transformer.norm.weight.grad)
        check(f"predictor[{variant}].g_nrmlp_B", mine.nrmlp.B.grad, ref.nrmlp.B.grad)
        names = ["transformer.layers.1.SpatialFFN1.norm2.weight", "EVT_Former.layers.0.temporal_MHSA.in_proj_weight",
                 "transformer.layers.0.EncDecAttn.out_proj.weight", "evt_posterior.conv2.0.weight"]
        rp, mp = dict(ref.named_parameters()), dict(mine.named_parameters())
        for n_ in names:
            check(f"predictor[{variant}].g[{n_}]", mp[n_].grad, rp[n_].grad)
        res.update(y_train=npy(yt), g_past=npy(p1.grad), g_tied_norm_w=npy(ref.transformer.norm.weight.grad),
                   g_tied_norm_b=npy(ref.transformer.norm.bias.grad), g_nrmlp_B=npy(ref.nrmlp.B.grad),
                   g_sffn1_norm2_w=npy(rp[names[0]].grad[::16]), g_evt_tmhsa_W_rows=npy(rp[names[1]].grad[::64]),
                   g_encdec_out_W_rows=npy(rp[names[2]].grad[::32]), g_post_conv2_w=npy(rp[names[3]].grad[::8, ::8]))
        # grad-norm of the decoder (what clip_grad_norm_ sees, ref Predictor.py:135)
        gn = torch.sqrt(sum((p.grad ** 2).sum() for p in ref.transformer.parameters()))
        gnm = torch.sqrt(sum((p.grad ** 2).sum() for p in mine.transformer.parameters()))
        check(f"predictor[{variant}].decoder_grad_norm", gnm, gn)
        res["decoder_grad_norm"] = npy(gn)
        save(f"predictor_{variant}", meta=np.array([N, To, Tp, 91, 92, 93, 94, 95]), **res)

        # ------------------------------------------------ training-step scalars (predictor-only flavour)
        ref2 = R.Predictor(8, 8, To + Tp, h, h, to, tp, 512, 'Add', 'layer', 256, 1, stochastic, 2,
                           norm=nn.LayerNorm(512), **kw)
        mine2 = oracle.Predictor(8, 8, To + Tp, h, h, to, tp, 512, 'Add', 'layer', 256, 1, stochastic, 2, **kw)
        O.key_hashed_fill(ref2, 101); O.key_hashed_fill(mine2, 101)
        if stochastic:
            mine2.evt_prior.eps_fn = mine2.evt_posterior.eps_fn = lambda shape: eps
        ref2.train(); mine2.train()
        opt_r = torch.optim.AdamW(ref2.parameters(), lr=1e-4)
        opt_m = torch.optim.AdamW(mine2.parameters(), lr=1e-4)
        steps = {}
        for it in range(2):
            # reference side: training_step_no_gan restated (features in, PF-L1 + KL)
            ref2.zero_grad()
            torch.randn = lambda *a, **k: eps
            try:
                o = ref2(past, fut) if stochastic else ref2(past)
            finally:
                torch.randn = real_randn
            pred = o[0] if stochastic else o
            kl = R.Div_KL(1e-6)(*o[1:]) if stochastic else torch.zeros(())
            pf = R.L1Loss(lam=0.01)(pred, fut)
            loss = pf + kl
            loss.backward()
            gn = torch.nn.utils.clip_grad_norm_(ref2.transformer.parameters(), max_norm=1.0, norm_type=2)
            opt_r.step()
            sm = oracle.predictor_train_step(mine2, opt_m, past, fut, 0.01, 1e-6, 1.0)
            # AdamW's first update is ~lr*sign(g): elements whose gradient is rounding noise
            # move by +-lr on either side, so quantities AFTER the first step agree only to ~1e-3.
            tol = 1e-5 if it == 0 else 2e-3
            check(f"train_step[{variant}][{it}].loss", torch.tensor(sm["loss"]), loss, tol)
            check(f"train_step[{variant}][{it}].grad_norm", torch.tensor(sm["grad_norm"]), gn, tol)
            steps[f"loss_{it}"] = npy(loss); steps[f"pf_{it}"] = npy(pf); steps[f"kl_{it}"] = npy(kl)
            steps[f"grad_norm_{it}"] = npy(gn)
            rp, mp = ref2.state_dict(), mine2.state_dict()
            for n_, key in (("w_dec_lin1", "transformer.layers.1.linear1.weight"), ("w_tied", "transformer.norm.weight"),
                            ("w_evt_fc1", "EVT_Former.layers.0.SpatialFFN.fc1.bias"), ("w_B", "nrmlp.B")):
                check(f"train_step[{variant}][{it}].{key}", mp[key], rp[key], tol)
                steps[f"{n_}_{it}"] = npy(rp[key].flatten()[:256])
        save(f"train_step_{variant}", meta=np.array([N, To, Tp, 101, 92, 93, 94]), **steps)

    # ------------------------------------------------------------------ full depth (4+8), D, eval: strided samples
    N, To, Tp = 1, 2, 3
    to, tp = torch.linspace(0, To - 1, To), torch.linspace(To, To + Tp - 1, Tp)
    ref = R.Predictor(8, 8, To + Tp, h, h, to, tp, 512, 'Add', 'layer', 256, 1, False, 8, norm=nn.LayerNorm(512),
                      evt_former=True, learn_evt_token=False, evt_former_num_layers=4, rand_context=False)
    mine = oracle.Predictor(8, 8, To + Tp, h, h, to, tp, 512, 'Add', 'layer', 256, 1, False, 8,
                            evt_former=True, learn_evt_token=False, evt_former_num_layers=4, rand_context=False)
    O.key_hashed_fill(ref, 111); O.key_hashed_fill(mine, 111)
    ref.eval(); mine.eval()
    past = O.synth_features((N, To, 512, 8, 8), 112)
    with torch.no_grad():
        y, ym = ref(past), mine(past)
    check("predictor_full_depth[D].eval", ym, y)
    save("predictor_full_D", y_strided=npy(y.flatten()[::7]), y_mean=npy(y.mean()), y_std=npy(y.std()),
         meta=np.array([N, To, Tp, 111, 112]))

    # ------------------------------------------------------------------ frozen autoencoder (stock-torch restatement
    # in npvp_amd/models/ResNetAutoEncoder.py) and the FULL Stage-2 step from pixels
    import npvp_amd
    for tag, (ci, ngf, nd, nres, S, out_layer) in {"64": (1, 64, 3, 2, 64, 'Sigmoid'), "128": (3, 32, 4, 3, 128, 'Tanh')}.items():
        re_ = R.ResnetEncoder(ci, ngf=ngf, n_downsampling=nd, num_res_blocks=nres, learn_3d=False)
        me = npvp_amd.ResnetEncoder(ci, ngf=ngf, n_downsampling=nd, num_res_blocks=nres, learn_3d=False)
        rd = R.ResnetDecoder(ci, ngf=ngf, n_downsampling=nd, out_layer=out_layer)
        md = npvp_amd.ResnetDecoder(ci, ngf=ngf, n_downsampling=nd, out_layer=out_layer)
        assert list(re_.state_dict().keys()) == list(me.state_dict().keys()) and list(rd.state_dict().keys()) == list(md.state_dict().keys())
        O.key_hashed_fill(re_, 121); O.key_hashed_fill(me, 121); O.key_hashed_fill(rd, 122); O.key_hashed_fill(md, 122)
        for m_ in (re_, me, rd, md):
            m_.eval()
        x = torch.rand(1, 2, ci, S, S, generator=torch.Generator().manual_seed(123))
        with torch.no_grad():
            fr, fm = re_(x), me(x)
        check(f"ae{tag}.enc", fm, fr)
        f1, f2 = fr.clone().requires_grad_(), fr.clone().requires_grad_()
        y1, y2 = rd(f1), md(f2)
        cot = O.seeded_randn(y1.shape, 124)
        (y1 * cot).sum().backward(); (y2 * cot).sum().backward()
        check(f"ae{tag}.dec", y2, y1); check(f"ae{tag}.dec_gin", f2.grad, f1.grad)
        save(f"ae_{tag}", feats=npy(fr), frames=npy(y1), g_feats=npy(f1.grad), meta=np.array([ci, ngf, nd, nres, S, 121, 122, 123, 124]))

    N, To, Tp = 2, 3, 4
    to, tp = torch.linspace(0, To - 1, To), torch.linspace(To, To + Tp - 1, Tp)
    kw = dict(evt_former=True, learn_evt_token=False, evt_former_num_layers=2, rand_context=False, dropout=0.0, drop_path=0.0)
    ref = R.Predictor(8, 8, To + Tp, h, h, to, tp, 512, 'Add', 'layer', 256, 1, True, 2, norm=nn.LayerNorm(512), **kw)
    mine = oracle.Predictor(8, 8, To + Tp, h, h, to, tp, 512, 'Add', 'layer', 256, 1, True, 2, **kw)
    O.key_hashed_fill(ref, 131); O.key_hashed_fill(mine, 131)
    renc = R.ResnetEncoder(1, ngf=64, n_downsampling=3, num_res_blocks=2, learn_3d=False)
    rdec = R.ResnetDecoder(1, ngf=64, n_downsampling=3, out_layer='Sigmoid')
    menc = npvp_amd.ResnetEncoder(1, ngf=64, n_downsampling=3, num_res_blocks=2, learn_3d=False)
    mdec = npvp_amd.ResnetDecoder(1, ngf=64, n_downsampling=3, out_layer='Sigmoid')
    for m_, sd in ((renc, 121), (menc, 121), (rdec, 122), (mdec, 122)):
        O.key_hashed_fill(m_, sd); m_.eval()
        for p_ in m_.parameters():
            p_.requires_grad_(False)
    g_ = torch.Generator().manual_seed(133)
    pf, ff = torch.rand(N, To, 1, 64, 64, generator=g_), torch.rand(N, Tp, 1, 64, 64, generator=g_)
    eps = O.seeded_randn((N, 512, 8, 8), 134)
    mine.evt_prior.eps_fn = mine.evt_posterior.eps_fn = lambda shape: eps
    ref.train(); mine.train()
    opt_r, opt_m = torch.optim.AdamW(ref.parameters(), lr=1e-4), torch.optim.AdamW(mine.parameters(), lr=1e-4)
    # reference side: shared_step + training_step_no_gan restated on the reference's own modules
    ref.zero_grad()
    with torch.no_grad():
        past_feats, fut_feats = renc(pf), renc(ff)
    torch.randn = lambda *a, **k: eps
    try:
        pred, mu_o, lv_o, mu_p, lv_p = ref(past_feats, fut_feats)
    finally:
        torch.randn = real_randn
    kl = R.Div_KL(1e-6)(mu_o, lv_o, mu_p, lv_p)
    frames = rdec(pred)
    img = R.L1Loss()(frames, ff); pfl = R.L1Loss(lam=0.01)(pred, fut_feats)
    loss = img + pfl + kl
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(ref.transformer.parameters(), max_norm=1.0, norm_type=2)
    opt_r.step()
    sm = oracle.full_train_step(mine, opt_m, menc, mdec, pf, ff, 0.01, 1e-6, 1.0)
    check("full_step.loss", torch.tensor(sm["loss"]), loss); check("full_step.Image_L1", torch.tensor(sm["Image_L1"]), img)
    check("full_step.grad_norm", torch.tensor(sm["grad_norm"]), gn)
    rp, mp = ref.state_dict(), mine.state_dict()
    check("full_step.w_dec_lin1", mp["transformer.layers.1.linear1.weight"], rp["transformer.layers.1.linear1.weight"])
    save("train_step_full_S", loss=npy(loss), img=npy(img), pf=npy(pfl), kl=npy(kl), grad_norm=npy(gn),
         w_dec_lin1=npy(rp["transformer.layers.1.linear1.weight"].flatten()[:256]),
         w_evt_fc1=npy(rp["EVT_Former.layers.0.SpatialFFN.fc1.bias"].flatten()[:256]), meta=np.array([N, To, Tp, 131, 121, 122, 133, 134]))

    # ------------------------------------------------------------------ unified model: random context
    N, T = 2, 7
    tl = torch.linspace(0, T - 1, T)
    kw = dict(evt_former=True, learn_evt_token=False, evt_former_num_layers=2, rand_context=True, dropout=0.0, drop_path=0.0)
    ref = R.Predictor(8, 8, T, h, h, tl[:3], tl[3:], 512, 'Add', 'layer', 256, 1, True, 2, norm=nn.LayerNorm(512), **kw)
    mine = oracle.Predictor(8, 8, T, h, h, tl[:3], tl[3:], 512, 'Add', 'layer', 256, 1, True, 2, **kw)
    idx_o, idx_p = torch.tensor([4, 0, 6, 2]), torch.tensor([5, 1, 3])          # unsorted on purpose
    # Fill seed: d/d nrmlp.B jumps by ~3e-3 when ONE NRMLP hidden unit changes sides of its ReLU, and fp32-grade arithmetic
    # paths differ by ~1e-6 in a pre-activation.  This case evaluates 448 coordinates x 3 x 512 units, too many for the 1e-5
    # margin of the stand-alone NRMLP vectors (seed 141 leaves a unit at 2.3e-6): take the first seed of 141, 1141, ... whose
    # closest unit is > 7e-6 from the kink (the search fills only the NRMLP, under its full-model state-dict keys).
    holder = nn.Module(); holder.nrmlp = ref.nrmlp
    best = (-1.0, 141)
    for seed_rc in range(141, 400141, 1000):
        O.key_hashed_fill(holder, seed_rc)
        with torch.no_grad():
            mg = min(nrmlp_margin(ref.nrmlp, ref.all_coor[idx_o, ...].flatten(0, 2)), nrmlp_margin(ref.nrmlp, ref.all_coor[idx_p, ...].flatten(0, 2)))
        best = max(best, (mg, seed_rc))
        if mg > 7e-6:
            break
    mg, seed_rc = best
    print(f"  randctx: fill seed {seed_rc}, smallest NRMLP |pre-activation| {mg:.2e}")
    O.key_hashed_fill(ref, seed_rc)
    O.key_hashed_fill(mine, seed_rc)
    clip = O.synth_features((N, T, 512, 8, 8), 142)
    batch = (clip[:, idx_o], clip[:, idx_p], idx_o, idx_p)
    # reference side of rand_context_batch_process (Predictor.py:241-251) on the reference module
    ref.observed_coor = ref.all_coor[idx_o, ...].flatten(0, 2)
    ref.predict_coor = ref.all_coor[idx_p, ...].flatten(0, 2)
    ref.TP = idx_p.shape[0]
    xo_m, xp_m = oracle.rand_context_batch_process(mine, batch)
    eps = O.seeded_randn((N, 512, 8, 8), 143)
    cot = O.seeded_randn((N, 3, 512, 8, 8), 144)
    mine.evt_prior.eps_fn = mine.evt_posterior.eps_fn = lambda shape: eps
    ref.train(); mine.train()
    xr = batch[0].clone().requires_grad_(); xm = xo_m.clone().requires_grad_()
    torch.randn = lambda *a, **k: eps
    try:
        yr = ref(xr, batch[1])
    finally:
        torch.randn = real_randn
    ym = mine(xm, xp_m)
    (yr[0] * yr[0] * cot).sum().backward(); (ym[0] * ym[0] * cot).sum().backward()
    check("randctx.out", ym[0], yr[0]); check("randctx.mu_p", ym[3], yr[3]); check("randctx.g_obs", xm.grad, xr.grad)
    check("randctx.gB", mine.nrmlp.B.grad, ref.nrmlp.B.grad)
    save("predictor_randctx_S", out=npy(yr[0]), mu_o=npy(yr[1]), mu_p=npy(yr[3]), g_obs=npy(xr.grad), gB=npy(ref.nrmlp.B.grad),
         g_tied=npy(ref.transformer.norm.weight.grad), meta=np.array([N, T, seed_rc, 142, 143, 144]))

    # ------------------------------------------------------------------ continuous time: reset_pos_coor with fractional
    # time-steps (ref Predictor.py:352-359, CoorGenerator submodules.py:339-366)
    N, T = 2, 7
    tl = torch.linspace(0, T - 1, T)
    kw = dict(evt_former=True, learn_evt_token=False, evt_former_num_layers=2, dropout=0.0, drop_path=0.0)
    ref = R.Predictor(8, 8, T, h, h, tl[:3], tl[3:], 512, 'SPADE', 'layer', 256, 1, False, 2, norm=nn.LayerNorm(512), **kw)
    mine = oracle.Predictor(8, 8, T, h, h, tl[:3], tl[3:], 512, 'SPADE', 'layer', 256, 1, False, 2, **kw)
    O.key_hashed_fill(ref, 151); O.key_hashed_fill(mine, 151)
    to_f, tp_f = torch.tensor([0.0, 1.5, 3.25]), torch.tensor([3.75, 4.5, 5.0, 6.5, 7.0])      # Tp changes from 4 to 5
    ref.reset_pos_coor(to_f, tp_f); mine.reset_pos_coor(to_f, tp_f)
    feats = O.synth_features((N, 3, 512, 8, 8), 152)
    cot = O.seeded_randn((N, 5, 512, 8, 8), 153)
    ref.train(); mine.train()
    xr = feats.clone().requires_grad_(); xm = feats.clone().requires_grad_()
    yr, ym = ref(xr), mine(xm)
    (yr * yr * cot).sum().backward(); (ym * ym * cot).sum().backward()
    check("fractime.out", ym, yr); check("fractime.g_obs", xm.grad, xr.grad); check("fractime.gB", mine.nrmlp.B.grad, ref.nrmlp.B.grad)
    save("predictor_fractime_D", out=npy(yr), g_obs=npy(xr.grad), gB=npy(ref.nrmlp.B.grad),
         coor_p=npy(ref.predict_coor), meta=np.array([N, T, 151, 152, 153]))

    # ------------------------------------------------------------------ evaluation metrics (ref utils/metrics.py)
    from oracle import metrics as OM
    RM = import_reference_metrics()
    arrays = {}
    for tag, shape, rng in (("g64", (3, 1, 64, 64), 1.0), ("rgb128", (2, 3, 128, 128), 255.0), ("odd", (2, 3, 45, 70), 1.0)):
        a = torch.rand(shape, generator=torch.Generator().manual_seed(161)) * rng
        b = (a + 0.1 * rng * O.seeded_randn(shape, 162)).clamp(0, rng)
        ps = RM.PSNR(a, b, data_range=rng, mean_flag=False); ms = RM.MSEScore(a, b, mean_flag=False)
        ss = RM.SSIM()(a / rng, b / rng, mean_flag=False)
        check(f"metrics.{tag}.psnr", OM.psnr_per_image(a, b, rng), ps); check(f"metrics.{tag}.mse", OM.mse_per_image(a, b), ms)
        check(f"metrics.{tag}.ssim", OM.ssim_per_image(a / rng, b / rng), ss)
        arrays.update({f"{tag}_psnr": ps.numpy(), f"{tag}_mse": ms.numpy(), f"{tag}_ssim": ss.numpy(),
                       f"{tag}_psnr_mean": np.float32(RM.PSNR(a, b, data_range=rng)), f"{tag}_ssim_mean": RM.SSIM()(a / rng, b / rng).numpy(),
                       f"{tag}_mse_mean": np.float32(RM.MSEScore(a, b))})
    ss7 = RM.SSIM(window_size=7)(a, b, mean_flag=False)
    check("metrics.odd.ssim_w7", OM.ssim_per_image(a, b, 7), ss7)
    arrays["odd_ssim_w7"] = ss7.numpy()

    class _Shift(nn.Module):            # a stand-in "model" with the call convention pred_ave_metrics expects
        def forward(self, past, fut, mask):
            return (fut * 0.9 + 0.05 * past[:, -1:],)
    loader = [(O.synth_features((2, 2, 1, 32, 32), 163 + i), O.synth_features((2, 3, 1, 32, 32), 173 + i)) for i in range(2)]
    renorm = lambda t: t * 0.5 + 0.25
    pam = RM.pred_ave_metrics(_Shift(), loader, RM.PSNR, renorm, 3, device='cpu')
    check("metrics.pred_ave_psnr", torch.tensor(OM.pred_ave_metrics(_Shift(), loader, lambda x, y: OM.psnr_per_image(x, y).mean(), renorm, 3)),
          torch.tensor(pam))
    arrays["pred_ave_psnr"] = pam
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **arrays)
    print("wrote metrics.npz")

    gen_val_steps(R)
    write_report()


if __name__ == "__main__":
    main()
