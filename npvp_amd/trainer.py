"""Stage-2 training step of the reference (LitPredictor.training_step_no_gan + shared_step +
configure_optimizers, ref/models/Predictor.py:124-148,172-194,196-218) without Lightning:

    zero_grad -> predictor(feats[, gt feats]) -> KL -> [frozen decoder -> image L1] -> feature L1
    -> backward -> clip_grad_norm_(predictor.transformer, 1.0) -> AdamW(lr 1e-4) -> cosine warm restarts

The clip + AdamW run as two kernels over FLAT parameter / gradient buffers (npvp_grad_norm_clip,
npvp_adamw_step); the flat gradient buffer is also what the data-parallel all-reduce buckets
(npvp_amd.dp) slice, so gradients are never copied.
"""
import math
import os

import torch
import yaml

from . import ops
from ._lib import lib, check
from .models.criterion import L1Loss, Div_KL


class FlatBuffers:
    """Re-point a module's parameters at slices of ONE flat fp32 buffer and their .grad at slices of a
    second one.  The parameters of `tail_module` are laid out last, so they form a contiguous range.
    Device agnostic (the data-parallel bucket logic is exercised on CPU/gloo); the optimiser kernels
    that consume the buffers are HIP only."""

    def __init__(self, module, tail_module=None):
        params = [p for p in module.parameters() if p.requires_grad]
        tail_ids = {id(p) for p in tail_module.parameters()} if tail_module is not None else set()
        ordered = [p for p in params if id(p) not in tail_ids] + [p for p in params if id(p) in tail_ids]
        dev = params[0].device
        pad = lambda n: (n + 3) // 4 * 4          # keep every slice 16-byte aligned
        total = sum(pad(p.numel()) for p in ordered)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.params, self.offsets, self.views = ordered, [], []
        off, tail_begin = 0, None
        with torch.no_grad():
            for p in ordered:
                n = p.numel()
                if tail_begin is None and id(p) in tail_ids:
                    tail_begin = off
                self.flat_p[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat_p[off:off + n].view(p.shape)
                ops.WeightPlanes.forget(p)               # planes cached for the storage p just left mirror dead memory
                p.grad = self.flat_g[off:off + n].view(p.shape)
                self.views.append(p.grad)
                p.__dict__["_npvp_flat"] = True          # ops.GradSink may accumulate into p.grad in place
                self.offsets.append((off, n))
                off += pad(n)
        self.total = total
        self.tail_begin = tail_begin if tail_begin is not None else total
        self.tail_end = total

    def zero_grad(self):
        # a backward pass that raised leaves queued weight-gradient closures and an open join behind (its engine callback never
        # ran): hand them over and join BEFORE the buffer is cleared, so that nothing stale lands in the new step's gradients and
        # the next backward registers its own callback
        if self.flat_g.is_cuda:
            ops.ReduceQueue.finish()
            ops.WgradStream.join()
        self.flat_g.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                p.grad = v

    def gather_stray_grads(self):
        """A caller that clears gradients with `module.zero_grad()` (set_to_none, as LitPredictor does, ref
        Predictor.py:126) detaches .grad from the flat buffer: autograd then allocates fresh gradient tensors.  Before
        the optimiser reads the flat buffer, copy such gradients into their slices (None -> zeros) and re-attach the
        views.  One identity check per parameter when nothing strayed."""
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is v:
                continue
            with torch.no_grad():
                if g is None:
                    v.zero_()
                else:
                    v.copy_(g)
            p.grad = v


class FlatAdamW:
    """torch.optim.AdamW semantics (lr, betas=(0.9,0.999), eps=1e-8, weight_decay=1e-2, ref Predictor.py:197)
    on one flat fp32 buffer (FlatBuffers).  The parameters of `clip_module` (the decoder
    `predictor.transformer`, ref :135 - which includes the LayerNorm it shares with the encoder) are laid
    out last so the clip_grad_norm_ range is contiguous."""

    def __init__(self, module, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, clip_module=None,
                 max_grad_norm=1.0, ctx=None):
        """ctx: the sched.StepContext this trainer's steps run under - dropout seed, gradient stream queue, deferred reductions,
        data-parallel listener, fp16 range guard.  Default: the context current at construction, i.e. the process's default one
        unless built under `ops.use(...)` (a single-trainer process: ops.rng.manual_seed etc. act on it).  Two trainers in one
        process each get their own with ctx=ops.StepContext(): their steps can then be interleaved without sharing any of it."""
        self.ctx = ctx if ctx is not None else ops.current()
        self.buf = FlatBuffers(module, clip_module)
        if not self.buf.flat_p.is_cuda:
            raise RuntimeError("FlatAdamW needs the module on an MI355X device (no CPU fallback)")
        dev = self.buf.flat_p.device
        self.flat_p, self.flat_g, self.total = self.buf.flat_p, self.buf.flat_g, self.buf.total
        self.params, self.offsets = self.buf.params, self.buf.offsets
        self.m = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.clip_begin, self.clip_end = (self.buf.tail_begin, self.buf.tail_end) if clip_module is not None else (0, 0)
        self.max_grad_norm = max_grad_norm
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.hyper = torch.tensor([lr, 0.0], dtype=torch.float32, device=dev)      # {lr, step}
        self.clip = torch.zeros(2, dtype=torch.float32, device=dev)                # {norm, coefficient}
        self._ws = torch.empty(1024, dtype=torch.float32, device=dev)
        self.param_groups = [{"lr": lr, "initial_lr": lr}]      # initial_lr: the base LR a torch scheduler would record at construction

    def zero_grad(self, set_to_none=False):
        with ops.use(self.ctx):
            self.buf.zero_grad()

    def set_lr(self, lr):
        self.hyper[0:1].fill_(lr)
        self.param_groups[0]["lr"] = lr

    def step(self):
        with ops.use(self.ctx):
            self._step()

    def _step(self):
        ops.ReduceQueue.finish()             # both: no-ops after a finished backward pass (the engine callbacks ran already)
        ops.WgradStream.join()
        self.buf.gather_stray_grads()
        L, P = lib(), ops._p
        s = P(torch.cuda.current_stream().cuda_stream)
        self.hyper[1:2].add_(1.0)
        clip_ptr = P(0)
        if self.clip_end > self.clip_begin and self.max_grad_norm is not None:
            g = self.flat_g[self.clip_begin:self.clip_end]
            check(L.npvp_grad_norm_clip(P(g.data_ptr()), g.numel(), float(self.max_grad_norm), P(self.clip.data_ptr()),
                                        P(self._ws.data_ptr()), self._ws.numel() * 4, s), "npvp_grad_norm_clip")
            clip_ptr = P(self.clip.data_ptr())
        check(L.npvp_adamw_step(P(self.flat_p.data_ptr()), P(self.flat_g.data_ptr()), P(self.m.data_ptr()),
                                P(self.v.data_ptr()), self.total, P(self.hyper.data_ptr()), self.betas[0], self.betas[1],
                                self.eps, self.weight_decay, clip_ptr, self.clip_begin, self.clip_end, 1, s),
              "npvp_adamw_step")
        ops.WeightPlanes.invalidate(self.flat_p)     # this trainer's parameters just changed under their cached planes

    def grad_norm(self):
        """Total L2 norm of the clipped range measured by the last step() (device scalar)."""
        return self.clip[0]

    # -- torch.optim.AdamW-format state, parameter order = `module.parameters()` (what LitPredictor's optimizer_P and a
    # -- Lightning checkpoint's optimizer_states[0] use, ref/models/Predictor.py:197)
    @staticmethod
    def _ref_layouts(module):
        """id(param) -> function(flat state slice) -> tensor in the reference's layout, for parameters this build
        stores differently from the reference (FrameLayerNorm keeps its (Ch,H,W) affine channels-last)."""
        from .models.VidHRFormer import FrameLayerNorm
        out = {}
        for mod in module.modules():
            if isinstance(mod, FrameLayerNorm):
                out[id(mod.weight)] = out[id(mod.bias)] = mod
        return out

    def state_dict(self, module):
        order = [p for p in module.parameters() if p.requires_grad]
        where = {id(p): self.offsets[i] for i, p in enumerate(self.params)}
        layouts = self._ref_layouts(module)
        step = torch.tensor(float(self.hyper[1]))
        state = {}
        for i, p in enumerate(order):
            off, n = where[id(p)]
            shape = (lambda t: layouts[id(p)].ref_view(t)) if id(p) in layouts else (lambda t: t.view(p.shape))
            state[i] = {"step": step.clone(), "exp_avg": shape(self.m[off:off + n]).detach().cpu().contiguous().clone(),
                        "exp_avg_sq": shape(self.v[off:off + n]).detach().cpu().contiguous().clone()}
        group = {"lr": self.param_groups[0]["lr"], "initial_lr": self.param_groups[0]["initial_lr"], "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(len(order)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd, module):
        order = [p for p in module.parameters() if p.requires_grad]
        where = {id(p): self.offsets[i] for i, p in enumerate(self.params)}
        g = sd["param_groups"][0]
        assert len(g["params"]) == len(order), "optimizer state does not match the module's parameter list"
        layouts = self._ref_layouts(module)
        t_is_ref = lambda t, mod: tuple(t.shape) == mod.normalized_shape
        step = 0.0
        with torch.no_grad():
            for i, p in enumerate(order):
                st = sd["state"].get(i, sd["state"].get(str(i)))
                if st is None:
                    continue
                off, n = where[id(p)]
                to_flat = (lambda t: t.permute(1, 2, 0).reshape(-1)) if (id(p) in layouts and t_is_ref(st["exp_avg"], layouts[id(p)])) \
                    else (lambda t: t.reshape(-1))
                self.m[off:off + n].copy_(to_flat(st["exp_avg"]))
                self.v[off:off + n].copy_(to_flat(st["exp_avg_sq"]))
                step = float(st["step"])
            self.hyper[1:2].fill_(step)
        self.set_lr(g["lr"])
        if "initial_lr" in g:               # (written by a torch scheduler into the group it was built on)
            self.param_groups[0]["initial_lr"] = g["initial_lr"]
        self.betas, self.eps, self.weight_decay = tuple(g["betas"]), g["eps"], g["weight_decay"]


def cosine_warm_restarts_lr(base_lr, eta_min, T_0, epoch_float):
    """torch CosineAnnealingWarmRestarts(T_0, T_mult=1, eta_min).step(epoch + batch_idx/len) as the reference
    calls it every iteration (ref/models/Predictor.py:144-148,213-215)."""
    t_cur = math.fmod(epoch_float, T_0)
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t_cur / T_0)) / 2


def build_predictor_from_cfg(cls, P, num_past, num_future, **overrides):
    """Construct a Predictor the way LitPredictor.__init__ does (ref/models/Predictor.py:28-47) from the
    `Predictor:` section of a reference YAML config."""
    h = torch.linspace(0, P['max_H'] - 1, P['max_H'])
    w = torch.linspace(0, P['max_W'] - 1, P['max_W'])
    to, tp = context_lists(P, num_past, num_future)
    assert P['max_T'] == num_past + num_future, "Incompatible max_T and clip length"
    return cls(P['max_H'], P['max_W'], P['max_T'], h, w, to, tp, P['embed_dim'], P['fuse_method'],
               P['param_free_norm_type'], P['evt_hidden_channels'], 1, P['stochastic'], P['transformer_layers'],
               evt_former=P['evt_former'], learn_evt_token=False, evt_former_num_layers=P['evt_former_num_layers'],
               rand_context=P['rand_context'], **overrides)


def load_config(path, batch_size=None, num_past=None, num_future=None):
    """Read a reference YAML (ref/configs/*.yaml; hydra is not needed: the reference uses the compose API
    with no overrides, ref/train_Predictor_lightning.py:51-56) and apply the benchmark's overrides of
    Dataset.batch_size / num_past_frames / num_future_frames / Predictor.max_T."""
    with open(path) as f:
        cfg = yaml.safe_load(f)
    D, P = cfg["Dataset"], cfg["Predictor"]
    if batch_size is not None:
        D["batch_size"] = batch_size
    if num_past is not None:
        D["num_past_frames"] = num_past
    if num_future is not None:
        D["num_future_frames"] = num_future
    P["max_T"] = D["num_past_frames"] + D["num_future_frames"]
    for k in ("predictor_lr", "scheduler_eta_min", "KL_beta", "lam_PF_L1", "max_grad_norm"):
        P[k] = float(P[k])
    return cfg


def predictor_train_step(predictor, opt, past_feats, future_feats, lam_PF_L1=0.01, KL_beta=1e-8, max_grad_norm=1.0,
                         frozen_dec=None, future_frames=None, sync=True, grad_sync=None):
    """One optimisation step on frozen-encoder features (predictor-only flavour), or the full step when the
    frozen decoder and the target frames are given.  `opt` must be a FlatAdamW.  With sync=False nothing is
    read back (bench / graph capture) and device scalars are returned."""
    with ops.use(opt.ctx):            # this trainer's dropout stream / gradient stream / deferred reductions / range guard
        return _predictor_train_step(predictor, opt, past_feats, future_feats, lam_PF_L1, KL_beta, max_grad_norm, frozen_dec,
                                     future_frames, sync, grad_sync)


def _predictor_train_step(predictor, opt, past_feats, future_feats, lam_PF_L1, KL_beta, max_grad_norm, frozen_dec, future_frames, sync,
                          grad_sync):
    dev = past_feats.device
    if grad_sync is not None and grad_sync.on and grad_sync.ctx is not opt.ctx:
        # (its listener would sit on another context's gradient sink - the in-place contributions of this step would go
        #  unreported and buckets would be all-reduced before their last write)
        raise RuntimeError("predictor_train_step: grad_sync listens on another scheduling context than the optimiser's; build it as "
                           "dp.GradSync(opt) (or pass ctx=opt.ctx)")
    ops.WgradStream.join()            # (a backward pass that raised leaves queued weight-gradient work and an open join behind)
    ops.rng.begin_step(dev)
    opt.max_grad_norm = max_grad_norm
    opt.zero_grad()
    if predictor.stochastic:
        pred, mu_o, lv_o, mu_p, lv_p = predictor(past_feats, future_feats)
        kl = Div_KL(KL_beta)(mu_o, lv_o, mu_p, lv_p)
    else:
        pred = predictor(past_feats)
        kl = torch.zeros((), dtype=torch.float32, device=dev)
    pf = L1Loss(lam=lam_PF_L1)(pred, future_feats)
    loss = pf + kl
    img = None
    if frozen_dec is not None:
        img = L1Loss()(frozen_dec(pred), future_frames)
        loss = loss + img
    loss.backward()
    if grad_sync is not None:
        grad_sync.finish()        # every bucket reduced (side stream) before the clip needs the gradients
    opt.step()
    out = {"loss": loss.detach(), "PF_L1": pf.detach(), "KL": kl.detach(), "grad_norm": opt.grad_norm(),
           "Image_L1": None if img is None else img.detach()}
    if sync:
        out = {k: (None if v is None else float(v)) for k, v in out.items()}
        # weight-gradient launches that met a feature 2^18 below its tensor's bound in this step (ops.RangeGuard: from the first
        # event on the process's weight gradients run in the bf16x6 arithmetic); read where the scalars are read anyway
        out["f16_range_events"] = ops.RangeGuard.poll(dev)
    return out


def full_train_step(predictor, opt, enc, dec, past_frames, future_frames, lam_PF_L1=0.01, KL_beta=1e-8, max_grad_norm=1.0,
                    sync=True, grad_sync=None):
    """The reference's complete Stage-2 step from pixels (shared_step + training_step_no_gan,
    ref/models/Predictor.py:124-148,172-194): frozen encoder on past and future frames under no_grad, predictor,
    frozen decoder on the predicted features (gradient flows through it), loss = L1(img) + lam*L1(feat) + KL."""
    enc.eval(); dec.eval()
    with torch.no_grad():
        past_feats = enc(past_frames)
        future_feats = enc(future_frames)
    return predictor_train_step(predictor, opt, past_feats, future_feats, lam_PF_L1, KL_beta, max_grad_norm,
                                frozen_dec=dec, future_frames=future_frames, sync=sync, grad_sync=grad_sync)


def predictor_val_step(predictor, past_feats, future_feats, lam_PF_L1=0.01, KL_beta=1e-8, frozen_dec=None, future_frames=None,
                       sync=True):
    """LitPredictor.validation_step + shared_step (ref/models/Predictor.py:150-170,172-194) on frozen-encoder features.
    Lightning runs validation with the module in eval mode under no_grad: dropout and drop-path are off, the EventEncoder's
    BatchNorm uses running statistics, and a stochastic predictor - still handed the ground-truth target features
    (shared_step :181-183) - runs both encoder passes, returns the 5-tuple and decodes from the PRIOR sample zo
    (ref :312-321).  Returns the scalars the reference logs as loss_val / PF_L1_val / KL_loss_val / Image_L1_val, and the
    prediction.  The module's train / eval state is restored."""
    dev = past_feats.device
    was_training = predictor.training
    predictor.eval()
    try:
        with torch.no_grad():
            if predictor.stochastic:
                pred, mu_o, lv_o, mu_p, lv_p = predictor(past_feats, future_feats)
                kl = Div_KL(KL_beta)(mu_o, lv_o, mu_p, lv_p)
            else:
                pred = predictor(past_feats)
                kl = torch.zeros((), dtype=torch.float32, device=dev)
            pf = L1Loss(lam=lam_PF_L1)(pred, future_feats)
            loss = pf + kl
            img = None
            if frozen_dec is not None:
                img = L1Loss()(frozen_dec(pred), future_frames)
                loss = loss + img
    finally:
        predictor.train(was_training)
    out = {"loss": loss, "PF_L1": pf, "KL": kl, "Image_L1": img}
    if sync:
        out = {k: (None if v is None else float(v)) for k, v in out.items()}
    out["pred"] = pred
    return out


def full_val_step(predictor, enc, dec, past_frames, future_frames, lam_PF_L1=0.01, KL_beta=1e-8, sync=True):
    """validation_step from pixels (ref/models/Predictor.py:150-170,172-194): frozen encoder, predictor (eval), frozen decoder."""
    enc.eval(); dec.eval()
    with torch.no_grad():
        past_feats = enc(past_frames)
        future_feats = enc(future_frames)
    return predictor_val_step(predictor, past_feats, future_feats, lam_PF_L1, KL_beta, frozen_dec=dec,
                              future_frames=future_frames, sync=sync)


# ---- batch shaping of the reference's other Stage-2 modes (ref/models/Predictor.py:30-40,62-70,241-262 and the
# ---- random-context collate ref/utils/dataset.py:162-178): SURVEY 8f "next" row #2
def context_lists(P, num_past, num_future):
    """(to_list, tp_list) as LitPredictor.__init__ derives them: VFI puts the context on both sides of the gap."""
    if P.get("VFI", False):
        cp, cf, nv = P["context_num_p"], P["context_num_f"], P["num_interpolate"]
        n = cp + cf + nv
        assert num_past + num_future == n, "Imcompatible VFI configurations"
        idx = torch.linspace(0, n - 1, n, dtype=torch.int64)
        return torch.cat([idx[0:cp], idx[-cf:]]), idx[cp:-cf]
    return (torch.linspace(0, num_past - 1, num_past),
            torch.linspace(num_past, num_past + num_future - 1, num_future))


def rand_context_collate(clip_batch, min_lo, max_lo, generator=None):
    """Split a batch of full clips (N,T,...) into a random context / target partition of the time axis: a random
    permutation of the T steps, the first `lo` (uniform in [min_lo, max_lo]) are observed, the rest predicted."""
    T = clip_batch.shape[1]
    perm = torch.randperm(T, generator=generator)
    lo = int(torch.randint(min_lo, max_lo + 1, (1,), generator=generator))
    idx_o, idx_p = perm[:lo], perm[lo:]
    return clip_batch[:, idx_o], clip_batch[:, idx_p], idx_o, idx_p


def rand_context_batch_process(predictor, batch):
    """Point the predictor's coordinate tables at this batch's context / target time-steps (ref :241-251)."""
    clip_o, clip_p, idx_o, idx_p = batch
    coor = predictor.all_coor
    predictor.observed_coor = coor[idx_o.to(coor.device)].flatten(0, 2)
    predictor.predict_coor = coor[idx_p.to(coor.device)].flatten(0, 2)
    predictor.TP = idx_p.shape[0]
    return clip_o, clip_p


def vfi_batch_process(batch, to_list, tp_list):
    """(past, future) -> (context frames on both sides, frames to interpolate)  (ref :253-259)"""
    clip = torch.cat(batch, dim=1)
    return clip[:, to_list], clip[:, tp_list]


# ---- checkpoint wire format of the reference (SURVEY 8f "next" row #3): a Lightning .ckpt of LitPredictor is a
# ---- torch.save'd dict {"state_dict": {...}, "optimizer_states": [...], "lr_schedulers": [...], "epoch", "global_step"}
# ---- whose state_dict keys carry the attribute prefixes predictor. / VPTR_Enc. / VPTR_Dec. (ref Predictor.py:17-19,43)
_CKPT_PREFIX = {"predictor": "predictor.", "enc": "VPTR_Enc.", "dec": "VPTR_Dec."}


def _reject_fused_autoencoder(module, role):
    """to_device_layout / fuse_frozen_autoencoder folds BatchNorm into the convolutions and renumbers the Sequentials: such a
    module no longer has the reference's VPTR_Enc. / VPTR_Dec. state-dict keys.  Load / save the pair BEFORE fusing it."""
    from .models.ResNetAutoEncoder import FoldedConvAct
    if any(isinstance(m, FoldedConvAct) for m in module.modules()):
        raise RuntimeError(f"{role}: this autoencoder has been fused for the device (to_device_layout): its state-dict keys are no "
                           "longer the reference's; save / load the checkpoint with the unfused modules, then call to_device_layout")


def save_lightning_checkpoint(path, predictor, enc=None, dec=None, opt=None, epoch=0, global_step=0, scheduler_T0=None,
                              scheduler_eta_min=1e-7):
    sd = {}
    for role, m in (("predictor", predictor), ("enc", enc), ("dec", dec)):
        if m is not None:
            _reject_fused_autoencoder(m, role)
            sd.update({_CKPT_PREFIX[role] + k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    ck = {"state_dict": sd, "epoch": int(epoch), "global_step": int(global_step), "pytorch-lightning_version": "1.6.5"}
    if opt is not None:
        ck["optimizer_states"] = [opt.state_dict(predictor)]
        # the full state_dict of torch's CosineAnnealingWarmRestarts (ref/models/Predictor.py:213-215)
        # The reference's scheduler steps once per EPOCH (configure_optimizers returns it with Lightning's default
        # interval='epoch', ref Predictor.py:213-215): T_cur / last_epoch count epochs; base_lrs = the LR the optimiser was
        # BUILT with; _step_count stays 0 (CosineAnnealingWarmRestarts.step overrides the base class's and never advances it -
        # checked against the live torch scheduler in tests/test_hip_golden.py).
        lr0 = opt.param_groups[0]["initial_lr"]
        ck["lr_schedulers"] = [{"T_0": scheduler_T0, "T_i": scheduler_T0, "T_mult": 1, "eta_min": scheduler_eta_min,
                                "T_cur": float(epoch % scheduler_T0), "base_lrs": [lr0], "last_epoch": float(epoch),
                                "_step_count": 0, "_last_lr": [opt.param_groups[0]["lr"]],
                                "_get_lr_called_within_step": False}] if scheduler_T0 else []
        ck["loops"] = None
    torch.save(ck, path)


def load_lightning_checkpoint(path, predictor, enc=None, dec=None, opt=None, strict=True, unsafe_pickle=False):
    """Load a reference Stage-2 checkpoint (or one written by save_lightning_checkpoint).  Returns (epoch, global_step).
    The file is read with torch.load(weights_only=True); unsafe_pickle=True falls back to full unpickling (arbitrary code
    execution: trusted files only)."""
    try:
        ck = torch.load(path, map_location="cpu", weights_only=True)       # tensors / containers only: no pickle code execution
    except Exception:
        if not unsafe_pickle:
            raise RuntimeError(f"{path}: not loadable with weights_only=True (a Lightning checkpoint may pickle hyper-parameter "
                               "objects); pass unsafe_pickle=True only for files you trust")
        ck = torch.load(path, map_location="cpu", weights_only=False)
    sd = ck["state_dict"]
    for role, m in (("predictor", predictor), ("enc", enc), ("dec", dec)):
        if m is None:
            continue
        _reject_fused_autoencoder(m, role)
        pre = _CKPT_PREFIX[role]
        part = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        if not part and not strict:
            continue
        m.load_state_dict(part, strict=strict)
    if opt is not None and ck.get("optimizer_states"):
        opt.load_state_dict(ck["optimizer_states"][0], predictor)
    return ck.get("epoch", 0), ck.get("global_step", 0)



_NODE_NAMES = ("kernel", "memcpy", "memset", "host", "child_graph", "empty", "wait_event", "event_record", "ext_sem_signal", "ext_sem_wait",
               "mem_alloc", "mem_free", "memcpy_from_symbol", "memcpy_to_symbol", "14", "other")


def _new_graph():
    """a torch graph object that KEEPS its hipGraph_t after the capture, so that the nodes can be counted (graph_census) before the
    explicit instantiation"""
    return torch.cuda.CUDAGraph(keep_graph=True)


def graph_census(graph):
    """{node type: count} of a captured torch.cuda.CUDAGraph(keep_graph=True) plus "memset_bytes" (sizes of its memset nodes) -
    include/npvp_hip.h npvp_graph_node_counts"""
    import ctypes
    counts, sets = (ctypes.c_longlong * 16)(), (ctypes.c_longlong * 64)()
    n = lib().npvp_graph_node_counts(graph.raw_cuda_graph(), ctypes.cast(counts, ctypes.c_void_p), ctypes.cast(sets, ctypes.c_void_p), 64)
    if n < 0:
        check(int(n), "npvp_graph_node_counts")
    out = {name: int(c) for name, c in zip(_NODE_NAMES, counts) if c}
    out["memset_bytes"] = [int(b) for b in sets[:min(64, int(counts[2]))]]
    return out


def _merge_census(total, one):
    for k, v in one.items():
        total[k] = total.get(k, []) + v if k == "memset_bytes" else total.get(k, 0) + v
    return total


def _require_memset_free(census, what):
    """Memset nodes and the runtime's replay modes (round 6, profiles/r06_graph_alloc_hazard.txt): a graph that ROCm 7.2 replays from
    the AQL packets it prepared at instantiation (the runtime's default, DEBUG_CLR_GRAPH_PACKET_CAPTURE=1) does not execute its memset
    nodes reliably: after other work of the process (an eager hipMemsetAsync of 1 MiB is enough: tools/graph_memset_node_repro.py) a
    memset node fills part of its buffer with a stale pattern or does nothing.  In the training step the hipMemsetAsync of the weight
    groups' amax table left garbage amaxes (wrong operand scales, wrong parameters) and the 4-byte semaphore of a torch reduction was
    not cleared (the loss scalar never written).  The step of this package therefore zero-fills with
    kernels and reduces with its own fixed-order sums, and a captured step that still contains a memset node (a caller's own ops
    inside the step, MIOpen) is refused in that replay mode instead of computing silently different numbers."""
    from . import graph_packet_capture
    n = census.get("memset", 0)
    if n and graph_packet_capture() and os.environ.get("NPVP_ALLOW_GRAPH_MEMSETS") != "1":      # (the switch: tools/graph_alloc_hazard.py)
        raise RuntimeError(f"{what}: the captured step contains {n} memset node(s) (bytes: {census['memset_bytes'][:8]}) and the HIP runtime "
                           "replays graphs from prepared packets (DEBUG_CLR_GRAPH_PACKET_CAPTURE is not 0): on ROCm 7.2 such a replay does "
                           "not execute memset nodes reliably (tools/graph_memset_node_repro.py).  Import npvp_amd before the first CUDA call without "
                           "NPVP_GRAPH_PACKET_CAPTURE=1 (it then selects the node-by-node replay mode), or remove the memsets from the step")


class StepTape:
    """A training step as a CHAIN of HIP-graph segments with host actions between them - how a step with collectives in it is
    replayed without the host enqueueing its ~1 000 launches: the kernels live in graphs, the collectives stay ordinary eager calls
    (no collective is ever captured; RCCL and gloo both work), and the host touches the step once per segment and action (~30 times).

    Recording: `begin()` starts a stream capture on the calling thread's current (non-default) stream; `cut(action)` - called from
    wherever the step wants a collective, also from the autograd engine's worker thread (hence capture mode "relaxed") - ends the
    current capture, notes the action, and begins the next capture on the same stream; `end()` closes the last one.  All segments
    allocate from ONE private pool, which is valid because they are always replayed in recording order.  Nothing executes while
    recording, so the actions are not run then either: the step's tensors hold garbage until the first replay.
    Every segment starts with a one-element tick kernel: a cut right after a cut must not leave an empty graph behind."""

    def __init__(self, dev):
        self.dev = dev
        self.items = []                  # torch.cuda.CUDAGraph | callable, in replay order
        self.pool = torch.cuda.graph_pool_handle()
        self._g = None
        self._tick = torch.zeros(1, dtype=torch.float32, device=dev)
        self.census = {}                 # node counts over all segments (graph_census)

    @property
    def recording(self):
        return self._g is not None

    @property
    def segments(self):
        return sum(1 for it in self.items if isinstance(it, torch.cuda.CUDAGraph))

    def begin(self):
        g = _new_graph()
        g.capture_begin(pool=self.pool, capture_error_mode="relaxed")
        self._g = g
        self._tick.add_(1.0)

    def _close(self, g):
        g.capture_end()
        _merge_census(self.census, graph_census(g))
        self.items.append(g)

    def cut(self, action):
        g, self._g = self._g, None
        self._close(g)
        self.items.append(action)
        self.begin()

    def end(self):
        """closes the last segment, checks the census (_require_memset_free) and instantiates every segment"""
        g, self._g = self._g, None
        if g is not None:
            self._close(g)
        _require_memset_free(self.census, "StepTape")
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.instantiate()

    def abort(self):
        """after an exception inside a recording: close the open capture (its graph is dropped)"""
        g, self._g = self._g, None
        if g is not None:
            try:
                g.capture_end()
            except Exception:
                pass
        self.items = []

    def replay(self):
        Graph = torch.cuda.CUDAGraph
        for it in self.items:
            if type(it) is Graph:
                it.replay()
            else:
                it()


class GraphedTrainStep:
    """The whole optimisation step (forward, losses, backward, clip, AdamW, weight-plane refresh) captured ONCE into a HIP graph and
    replayed per step: the host side of a step (1 000 - 3 000 launches and their Python at any batch size) shrinks to one graph
    launch, which is what small per-GPU batches (c0: 4 clips, c3 / c4: 8 clips per GPU) are bound by on a slow host.  Everything a
    replay must see differently lives in device memory: the input batch (copied into static buffers), the dropout seed (bumped by a
    kernel inside the graph), lr / step count / clip coefficient.

    single_stream (default): the step is captured WITHOUT the gradient stream - a graph with cross-stream edges replays at twice the
    time of a single chain of nodes on ROCm 7.2 (56 against 31 ms for an 8-clip step, in either replay mode of the runtime).  The
    overlap the second stream gives the eager step comes from inside the launches instead (grouped GEMM launches).

    Replay mode of the runtime and memset nodes (round 6, profiles/r06_graph_alloc_hazard.txt): ROCm 7.2 replays a graph from AQL
    packets it prepared at instantiation (0.3 - 1.5 ms of host per replay of ~1 000 kernels) and in that mode does NOT execute the
    graph's memset nodes reliably (stale fill patterns after other work of the process: tools/graph_memset_node_repro.py); `import npvp_amd` therefore selects DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 - the
    runtime marshals every node at launch (2.5 - 8 ms of host per replay), which is exact with any step.  The step of this package
    contains no memset node (zero fills are kernels, the losses are the library's fixed-order sums), `census` says so for every
    capture, and with the prepared-packet mode switched on (NPVP_GRAPH_PACKET_CAPTURE=1) a capture that does contain one - a caller's
    own ops, MIOpen - is refused (_require_memset_free).  A memset-free step is equal to the bit in both modes and, being bound by
    its kernels, equally fast (c4 shard 34.0 / 34.1 ms, c2 239.0 / 238.6 ms on one box).

    Range of the fp16 arithmetic (ops.RangeGuard): the weight-gradient kernels baked into the graph raise the device counter like the
    eager ones; every `poll_every` replays the counter is copied to pinned host memory WITHOUT blocking and looked at on the next call.
    From the first event on the step is captured again with the weight gradients in the bf16x6 arithmetic (`recaptures` counts them;
    `range_events` is the running total) - a replayed graph cannot be switched, only re-captured.

    Data parallel (grad_sync = the trainer's dp.GradSync): collectives are NOT captured.  The step is recorded as a StepTape - a
    chain of single-stream graph segments cut wherever the eager step issues a collective (each gradient bucket's all-reduce, the
    EventEncoder's SyncBatchNorm statistics forward and backward, the closing wait before the clip) - and a replay alternates graph
    launches with those collectives, issued eagerly exactly as the eager step does (bucket all-reduces on GradSync's side stream
    behind an event of the compute stream, so they still overlap the rest of the backward pass).  ~15 segments + ~15 collectives
    per step instead of ~1 000 enqueued launches: the data-parallel shard step no longer depends on the host's speed.  The warm-up
    (at least two eager steps, in the single-stream schedule the capture uses) teaches GradSync the contribution counts."""

    def __init__(self, predictor, opt, past_feats, future_feats, lam_PF_L1=0.01, KL_beta=1e-8, max_grad_norm=1.0, warmup=3,
                 single_stream=True, poll_every=32, grad_sync=None):
        """warmup: eager optimiser steps taken on (past_feats, future_feats) BEFORE the capture - they are real steps (parameters,
        Adam state, step count and dropout seed advance `warmup` times); pass warmup=0 when the caller has already stepped the model
        eagerly on this device (then nothing but the capture itself happens, and the capture executes nothing)."""
        self.opt = opt
        self.past, self.fut = past_feats.clone(), future_feats.clone()
        self._args = (predictor, opt, self.past, self.fut, lam_PF_L1, KL_beta, max_grad_norm)
        self.grad_sync = grad_sync if (grad_sync is not None and grad_sync.on) else None
        if self.grad_sync is not None:
            assert single_stream, "a data-parallel step is recorded single-stream (graph segments are chains)"
            warmup = max(2, warmup)
        self.tape = None
        self.census = {}                    # node counts of the captured step (graph_census); census.get("memset", 0) == 0 for this package's step
        self.single_stream, self.poll_every, self.warmup = single_stream, max(1, int(poll_every)), warmup
        self.replays = self.recaptures = self.range_events = 0
        self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._flag_event = None
        self._capture(self.warmup)

    def _capture(self, warmup=0):
        """warmup: eager steps before the capture - REAL optimiser steps on the batch in the static buffers (the construction's
        `warmup`, documented there: lazy streams / workspaces / weight planes get created).  A re-capture (range event) runs none:
        everything lazy exists already, and extra steps would apply extra AdamW updates, advance the step count and the dropout
        seed behind the caller's back - the trajectory of a replayed run must be the eager one's."""
        with ops.use(self.opt.ctx):
            self._capture_in_ctx(warmup)

    def _capture_in_ctx(self, warmup):
        dev = self.past.device
        two_streams = ops.WgradStream.enabled
        if self.single_stream:
            ops.WgradStream.join()
            ops.WgradStream.enabled = False
        gs = self.grad_sync
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):               # eager warm-up on a side stream (lazy streams / workspaces get created)
                if gs is not None and warmup:
                    gs.relearn()                        # (the contribution counts of THIS schedule: the first warm-up step learns them)
                for _ in range(warmup):
                    predictor_train_step(*self._args, sync=False, grad_sync=gs)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            # the eager steps' cached blocks go back to the device: the capture allocates the step's whole working set again, from its
            # private pool (c2: 104 GiB beside an eager pool that an un-synchronised warm-up grows to 200 GiB would not fit in 288)
            torch.cuda.empty_cache()
            # a live probe (bench.py) brackets launches with library events that the capture carries (sched.ProbeEvent): the records
            # of the warm-up's eager launches are dropped, the capture's are re-stamped by every replay
            if ops.GemmProbe.armed:
                ops.GemmProbe.records = []
            if ops.HbmProbe.armed:
                ops.HbmProbe.records = []
            # amax slots (f16x3 GEMMs) must be zero when their tensor is produced: the captured step cuts its slots from chunks
            # created INSIDE the capture, so their zero fill is a node of the graph and every replay starts from clean slots
            ops.AmaxSlot.reset_chunks()
            n0 = lib().npvp_launch_count()
            if gs is None:
                self.graph = _new_graph()
                with torch.cuda.graph(self.graph):
                    self.out = predictor_train_step(*self._args, sync=False)
                self.census = graph_census(self.graph)
                _require_memset_free(self.census, "GraphedTrainStep")
                self.graph.instantiate()
            else:
                self._record_segments(dev, gs)
                self.census = dict(self.tape.census)
            if os.environ.get("NPVP_GRAPH_NODES") == "1":       # (diagnosis)
                print(f"[GraphedTrainStep] nodes of the captured step: {self.census}", flush=True)
            self.launches = lib().npvp_launch_count() - n0          # library launches of one step (what a replay enqueues on the device)
            # the graph has the addresses of the weight-plane tables and amax slots baked in: they live as long as the graph does
            self._plane_tables = list(ops.WeightPlanes._tables or [])
            # ... and so do the probe events whose record nodes the capture carries (a destroyed event under a replay is a crash)
            self._probe_events = (list(ops.GemmProbe.records) if ops.GemmProbe.armed else []) + (list(ops.HbmProbe.records) if ops.HbmProbe.armed else [])
        finally:
            ops.AmaxSlot.reset_chunks()         # (the graph's private-pool chunk is not for eager code - also after a failed capture)
            ops.WgradStream.enabled = two_streams

    def _record_segments(self, dev, gs):
        """the data-parallel step as a StepTape (see the class docstring): dp routes every collective to the tape while it records"""
        from . import dp
        if gs.expected is None:
            raise RuntimeError("GraphedTrainStep(grad_sync=...): GradSync has not learned its contribution counts - run at least two "
                               "eager steps (warmup >= 2) before the capture")
        tape = StepTape(dev)
        cs = torch.cuda.Stream(device=dev)
        cs.wait_stream(torch.cuda.current_stream(dev))
        prev = dp.set_tape(tape)
        try:
            with torch.cuda.stream(cs):
                tape.begin()
                self.out = predictor_train_step(*self._args, sync=False, grad_sync=gs)
                tape.end()
        except BaseException:
            tape.abort()
            raise
        finally:
            dp.set_tape(prev)
        torch.cuda.current_stream(dev).wait_stream(cs)
        self.tape, self.graph = tape, None

    def _poll_range(self):
        """non-blocking: look at the copy issued `poll_every` replays ago; issue the next one"""
        with ops.use(self.opt.ctx):
            self._poll_range_in_ctx()

    def _poll_range_in_ctx(self):
        dev = self.past.device
        if self._flag_event is not None and self._flag_event.query():
            n = int(self._flag_host[0])
            self._flag_event = None
            if n:
                flag = ops.RangeGuard.flag(dev)
                flag.zero_()
                ops.RangeGuard.events += n
                self.range_events += n
                if not ops.RangeGuard.fallback:
                    ops.RangeGuard.fallback = True      # the new capture bakes the bf16x6 weight gradients in
                    self.recaptures += 1
                    self._capture(0)
        if self._flag_event is None and self.replays % self.poll_every == 0:
            flag = ops.RangeGuard._flags.get(dev)
            if flag is not None:
                self._flag_host.copy_(flag, non_blocking=True)
                self._flag_event = torch.cuda.Event()
                self._flag_event.record()

    def __call__(self, past_feats=None, future_feats=None, lr=None):
        if lr is not None:
            self.opt.set_lr(lr)
        if past_feats is not None:
            self.past.copy_(past_feats)
        if future_feats is not None:
            self.fut.copy_(future_feats)
        self._poll_range()                  # (before the replay: a re-capture replaces self.out)
        ops.WeightPlanes.refresh_if_stale()
        if self.tape is not None:
            self.tape.replay()
        else:
            self.graph.replay()
        self.replays += 1
        return self.out
