"""GPU: the HIP path (npvp_amd, through the C ABI) reproduces the golden vectors captured from the
imported reference, and agrees with the oracle at larger / full BASELINE sizes.  Bar: 1e-3 rel fp32
(BASELINE.json north_star); the fp32-MFMA GEMM path is held to 1e-4 here so regressions show early."""
import math
import os

import pytest
import torch

import golden_cases as GC
from oracle import ops as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


MODE = "f16x3"
DEFAULT_MODE = "f16x3"
MODE_TOL = {"f32": 1e-4, "f16x3": 1e-4, "bf16x6": 1e-4, "bf16x3": 1e-2}       # bf16x3 (NPVP_TEST_ALL_MODES only): 2^-16 per product in EVERY GEMM - parameter gradients at the end of the longest chains measure 5.9e-3 .. 6.3e-3


@pytest.fixture(scope="module", params=["f16x3", "f32"] + (["bf16x6", "bf16x3"] if os.environ.get("NPVP_TEST_ALL_MODES") else []))
def impl(request):
    """The exact fp32-MFMA path and the default path (f16x3: two-term fp16 split on the large GEMMs, three-term bf16 split on
    the small ones) must reproduce the reference's vectors to 1e-4 (north_star bar: 1e-3); NPVP_TEST_ALL_MODES adds the pure
    bf16x6 mode and bf16x3 (2-term bf16 split, ~2^-16 product error, an opt-in fast mode for weight gradients: forward
    outputs meet 1e-3 but the deepest gradients (d/d input features, d/d nrmlp.B) only reach ~3e-3, so it is held to 5e-3)."""
    import npvp_amd
    global TOL, MODE
    assert torch.cuda.is_available()
    npvp_amd.ops.set_gemm_precision(request.param)
    TOL, MODE = MODE_TOL[request.param], request.param
    yield npvp_amd
    npvp_amd.ops.set_gemm_precision(DEFAULT_MODE)
    TOL = 1e-4


@pytest.mark.parametrize("norm", ["layer", "instance"])
def test_posfuse(impl, norm):
    GC.compare(GC.case_posfuse(impl, DEV, norm), GC.load(f"posfuse_{norm}"), TOL, tag=f"posfuse_{norm}[{MODE}]")


def test_decoder_block_with_instance_fuser_against_oracle(impl):
    """PosFeatFuser('instance') inside a decoder block (the blocks then run their per-kernel autograd path instead of the
    sub-layer nodes): forward and input gradients against the oracle."""
    import oracle
    N, T2, T1 = 2, 3, 2

    def run(mod_impl, dev):
        m = mod_impl.VidHRFormerBlockDecNAR(8, 8, 512, 8, 4, 0.0, 0.0, 4, 1024)
        O.key_hashed_fill(m, 61)
        m = m.to(dev).train()
        tgt = (0.3 * O.seeded_randn((N, T2, 8, 8, 512), 62)).to(dev).requires_grad_()
        qe = (0.5 * O.seeded_randn((N, 8, 8, 512), 63)).to(dev).requires_grad_()
        mem = O.synth_features((N, T1, 8, 8, 512), 64).to(dev).requires_grad_()
        mb, mg = GC.pos_tables(T1, 65, True, dev)
        tb, tg = GC.pos_tables(T2, 66, True, dev)
        cot = O.seeded_randn((N, T2, 8, 8, 512), 67).to(dev)
        y = m(tgt, qe, mem, (mb, mg), (tb, tg), mod_impl.PosFeatFuser(512, 'instance'))
        # (norm5.weight, not .bias: an instance norm removes any per-channel constant, so d/d norm5.bias is pure rounding noise)
        g = torch.autograd.grad((y * cot).sum(), [tgt, qe, mem, m.norm5.weight, m.norm1.bias])
        return [t.detach().cpu() for t in (y, *g)]

    want, got = run(oracle, "cpu"), run(impl, DEV)
    for a, b, n in zip(got, want, ["y", "g_tgt", "g_qe", "g_mem", "g_norm5_w", "g_norm1_b"]):
        e = GC.rel_err(a, b)
        assert e < TOL, f"{n}: {e:.3e}"


@pytest.mark.parametrize("fuse", ["Add", "SPADE"])
def test_nrmlp(impl, fuse):
    GC.compare(GC.case_nrmlp(impl, DEV, fuse), GC.load(f"nrmlp_{fuse}"), TOL, tag=f"nrmlp_{fuse}[{MODE}]")


def test_slmhsa(impl):
    GC.compare(GC.case_slmhsa(impl, DEV), GC.load("slmhsa"), TOL, tag=f"slmhsa[{MODE}]")


def test_mlpdwbn(impl):
    GC.compare(GC.case_mlpdwbn(impl, DEV), GC.load("mlpdwbn"), TOL, tag=f"mlpdwbn[{MODE}]")


def test_block_enc_mask_quirk(impl):
    GC.compare(GC.case_block_enc(impl, DEV), GC.load("block_enc"), TOL, tag=f"block_enc[{MODE}]")


def test_block_dec(impl):
    GC.compare(GC.case_block_dec(impl, DEV), GC.load("block_dec"), TOL, tag=f"block_dec[{MODE}]")


def test_evtenc(impl):
    GC.compare(GC.case_evtenc(impl, DEV), GC.load("evtenc"), TOL, tag=f"evtenc[{MODE}]")


def test_losses(impl):
    GC.compare(GC.case_losses(impl, DEV), GC.load("losses"), TOL, tag=f"losses[{MODE}]")


@pytest.mark.parametrize("variant", ["D", "S"])
def test_predictor(impl, variant):
    GC.compare(GC.case_predictor(impl, DEV, variant), GC.load(f"predictor_{variant}"), TOL, tag=f"predictor_{variant}[{MODE}]")


@pytest.mark.parametrize("variant", ["D", "S"])
def test_train_step(impl, variant):
    mk = lambda m: impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
    res = GC.case_train_step(impl, DEV, variant, make_opt=mk)
    g = GC.load(f"train_step_{variant}")
    GC.compare({k: v for k, v in res.items() if k.endswith("_0")}, {k: v for k, v in g.items() if k.endswith("_0")}, TOL)
    GC.compare({k: v for k, v in res.items() if k.endswith("_1")}, {k: v for k, v in g.items() if k.endswith("_1")}, max(3e-3, TOL))


@pytest.mark.parametrize("path", ["stock", "fused"])
@pytest.mark.parametrize("tag", ["64", "128"])
def test_frozen_autoencoder(impl, tag, path):
    """frozen AE vs the reference's vectors.  stock: the modules as built on PyTorch-ROCm (MIOpen); fused: to_device_layout =
    channels_last encoder, BatchNorm folded into the convolutions, bias / ReLU / skip-add / tanh in csrc/ae.hip's epilogue pass,
    decoder input gradient through npvp_act_bwd"""
    if MODE != DEFAULT_MODE:
        pytest.skip("the frozen autoencoder has no GEMM of the predictor's kind: one arithmetic mode is enough")
    res = GC.case_ae(impl, DEV, tag, device_layout=impl.to_device_layout if path == "fused" else None)
    gold = GC.load(f"ae_{tag}")
    # forward: MIOpen convolutions vs the CPU reference.  The decoder's input gradient passes through its ReLU masks:
    # a handful of units within rounding noise of 0 flip with MIOpen's algorithm choice (run-to-run), each flip a finite
    # gradient change - observed 1e-5 .. 1.2e-3 rel-L2 on the 128x128 decoder, so that key gets its own bound.
    g_res, g_gold = res.pop("g_feats"), gold.pop("g_feats")
    GC.compare(res, gold, 1e-4, tag=f"ae_{tag}[{MODE}]")
    GC.compare({"g_feats": g_res}, {"g_feats": g_gold}, 5e-3, tag=f"ae_{tag}.g_feats[{MODE}]")


@pytest.mark.parametrize("layout", ["nchw", "encoder_channels_last"])
def test_full_step_from_pixels(impl, layout):
    """layout = encoder_channels_last: the frozen encoder runs NHWC and hands the predictor its canonical layout directly"""
    mk = lambda m: impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
    dl = impl.to_device_layout if layout == "encoder_channels_last" else None
    res, gold = GC.case_full_step(impl, impl, DEV, make_opt=mk, device_layout=dl), GC.load("train_step_full_S")
    # losses: forward only.  grad_norm and the updated weights hang on the frozen decoder's INPUT GRADIENT, which torch / MIOpen
    # computes: a few ReLU units within rounding noise of 0 flip with MIOpen's algorithm choice from run to run (see
    # test_frozen_autoencoder) - the same process repeating this case gives grad_norm 4.2e-5, 9.4e-5 or 1.2e-4 off the vector
    # (tools/f32_repeat.py; the predictor-only step repeats bit for bit, tools/f32_trace.py), so those keys get that bound
    via_decoder = ("grad_norm", "w_dec_lin1", "w_evt_fc1")
    GC.compare({k: v for k, v in res.items() if k not in via_decoder}, {k: v for k, v in gold.items() if k not in via_decoder}, TOL,
               tag=f"full_step[{MODE},{layout}]")
    GC.compare({k: res[k] for k in via_decoder}, {k: gold[k] for k in via_decoder}, 1e-3, tag=f"full_step.grads[{MODE},{layout}]")


@pytest.mark.parametrize("layout", ["nchw", "encoder_channels_last"])
@pytest.mark.parametrize("variant", ["D", "S"])
def test_validation_step(impl, variant, layout):
    """LitPredictor.validation_step (ref Predictor.py:150-170,172-194) against the reference's vectors: eval mode under no_grad
    (dropout / drop-path 0.1 configured but inactive, BatchNorm on running statistics), NPVP-S handed the ground truth ->
    5-tuple, prediction decoded from the PRIOR sample zo (ref :312-321), the *_val losses; the module's train mode restored."""
    dl = impl.to_device_layout if layout == "encoder_channels_last" else None
    GC.compare(GC.case_val_step(impl, impl, DEV, variant, device_layout=dl), GC.load(f"val_step_{variant}"), TOL,
               tag=f"val_step_{variant}[{MODE},{layout}]")


def test_bf16x6_mode_stays_in_the_default_session(impl):
    """the six-term bf16 arithmetic is a selectable mode (bench.py --gemm bf16x6, NPVP_GEMM) and what every shape the
    fp16 kernels do not take falls back to; its wide kernels (bf16 weight planes, gemm_wide_kernel) must not go untested when
    NPVP_TEST_ALL_MODES is unset: one whole-predictor case and one training step per default session."""
    import npvp_amd
    if MODE != DEFAULT_MODE:
        pytest.skip("runs once, beside the default mode")
    npvp_amd.ops.set_gemm_precision("bf16x6")
    try:
        GC.compare(GC.case_predictor(impl, DEV, "D"), GC.load("predictor_D"), 1e-4, tag="predictor_D[bf16x6]")
        mk = lambda m: impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
        res, g = GC.case_train_step(impl, DEV, "S", make_opt=mk), GC.load("train_step_S")
        GC.compare({k: v for k, v in res.items() if k.endswith("_0")}, {k: v for k, v in g.items() if k.endswith("_0")}, 1e-4,
                   tag="train_step_S[bf16x6]")
    finally:
        npvp_amd.ops.set_gemm_precision(MODE)


def test_backward_that_raises_leaves_no_stale_gradient_work(impl):
    """WgradStream queues closures; a backward pass that raises never runs the engine callback that joins the gradient
    stream.  FlatAdamW.zero_grad / step join first: the next step's gradients equal those of a fresh run."""
    from npvp_amd import ops
    N, To, Tp = 2, 3, 4
    past = O.synth_features((N, To, 512, 8, 8), 92).to(DEV)
    fut = O.synth_features((N, Tp, 512, 8, 8), 93).to(DEV)

    def grads(poison):
        m = GC._small_predictor(impl, False, 101, DEV)
        m.train()
        opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
        if poison:
            opt.zero_grad()

            class Boom(torch.autograd.Function):
                @staticmethod
                def forward(ctx, x):
                    return x.clone()

                @staticmethod
                def backward(ctx, g):
                    raise RuntimeError("boom")
            # the failing node sits at the INPUT: the whole predictor's backward (weight-gradient closures queued on the way) runs first
            y = m(Boom.apply(past.clone().requires_grad_()))
            with pytest.raises(RuntimeError, match="boom"):
                (y * y).sum().backward()
            assert ops.WgradStream._pending is not None, "expected an open gradient-stream join after the failed backward"
        opt.zero_grad()
        (m(past) - fut).abs().mean().backward()
        ops.WgradStream.join()
        torch.cuda.synchronize()
        return opt.flat_g.clone()

    a, b = grads(False), grads(True)
    assert GC.rel_err(b, a) < 1e-6, "stale gradient-stream work leaked into the step after a failed backward"


def test_chained_split_k_reductions_are_bit_identical(impl):
    """ops.WgradChain: a weight gradient's split-K reduction done by extra workgroups of the NEXT weight-gradient launch (or by
    npvp_splitk_reduce_job at the end of backward) sums the same slabs in the same order as the stand-alone reduction launch:
    the whole flat gradient of a step must be bit-identical with the chain on and off (4 096 decoder token rows: the fp16
    weight-gradient kernel with split-K; dropout on, so the row-group masks ride in the chained launches too)."""
    from npvp_amd import ops
    if MODE != "f16x3":
        pytest.skip("the chain belongs to the fp16 weight-gradient kernel")
    N, To, Tp = 8, 2, 8
    past = O.synth_features((N, To, 512, 8, 8), 92).to(DEV)
    fut = O.synth_features((N, Tp, 512, 8, 8), 93).to(DEV)
    old, old_fused = ops.WgradChain.enabled, ops.FusedLinearBwd.enabled
    ops.FusedLinearBwd.enabled = False          # (the fused dgrad + weight-gradient launches hand their reductions to ReduceQueue)
    flat, launches = {}, {}
    try:
        for on in (True, False):
            ops.WgradChain.enabled = on
            m = GC._small_predictor(impl, False, 101, DEV, To=To, Tp=Tp, dropout=0.1, drop_path=0.1)
            m.train()
            opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
            ops.rng.manual_seed(777, torch.device(DEV))
            ops.rng.begin_step(torch.device(DEV))
            opt.zero_grad()
            (m(past) - fut).abs().mean().backward()
            ops.WgradStream.join()
            torch.cuda.synchronize()
            assert not ops.WgradChain._pending, "a deferred reduction was left behind after the join"
            flat[on] = opt.flat_g.clone()
        assert float(flat[True].abs().max()) > 0
        assert torch.equal(flat[True], flat[False]), f"chained vs stand-alone reductions differ: {GC.rel_err(flat[True], flat[False]):.3e}"
    finally:
        ops.WgradChain.enabled, ops.FusedLinearBwd.enabled = old, old_fused


@pytest.mark.parametrize("two_streams", [True, False])
def test_fused_linear_backward_is_bit_identical(impl, two_streams):
    """ops.linear_bwd: the dgrad and the weight gradient of a linear layer in ONE launch (npvp_linear_bwd_f16: the same two kernel
    bodies on disjoint workgroups, the split-K reduction queued for the gradient stream - or, single stream, riding in the next fused
    launch) against the two launches on two streams.  Same tiles, same split counts, same summation order: the flat gradient of a
    whole step (4 096 decoder token rows, dropout and drop-path on) must be bit-identical, with and without a gradient stream."""
    from npvp_amd import ops
    if MODE != "f16x3":
        pytest.skip("the fused launch belongs to the fp16 kernels")
    N, To, Tp = 8, 2, 8
    past = O.synth_features((N, To, 512, 8, 8), 92).to(DEV)
    fut = O.synth_features((N, Tp, 512, 8, 8), 93).to(DEV)
    old = (ops.FusedLinearBwd.enabled, ops.WgradStream.enabled, ops.FusedLinearBwd.with_gradient_stream)
    flat, launches = {}, {}
    try:
        ops.WgradStream.join()
        ops.WgradStream.enabled = two_streams
        ops.FusedLinearBwd.with_gradient_stream = True      # by default only the step without a gradient stream fuses
        for on in (True, False):
            ops.FusedLinearBwd.enabled = on
            m = GC._small_predictor(impl, False, 101, DEV, To=To, Tp=Tp, dropout=0.1, drop_path=0.1)
            m.train()
            opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
            ops.rng.manual_seed(777, torch.device(DEV))
            ops.rng.begin_step(torch.device(DEV))
            opt.zero_grad()
            n0 = impl._lib.lib().npvp_launch_count()
            (m(past) - fut).abs().mean().backward()
            ops.WgradStream.join()
            torch.cuda.synchronize()
            launches[on] = impl._lib.lib().npvp_launch_count() - n0
            assert not ops.WgradChain._pending and not ops.ReduceQueue.pending(), "a deferred reduction was left behind"
            flat[on] = opt.flat_g.clone()
        assert float(flat[True].abs().max()) > 0
        assert launches[True] < launches[False], launches
        assert torch.equal(flat[True], flat[False]), f"fused vs two launches: {GC.rel_err(flat[True], flat[False]):.3e}"
    finally:
        ops.FusedLinearBwd.enabled, ops.WgradStream.enabled, ops.FusedLinearBwd.with_gradient_stream = old


def test_two_trainers_in_one_process_do_not_share_state(impl):
    """npvp_amd.sched.StepContext: the dropout seed and salts, the gradient-stream queue, the deferred reductions, the range guard
    and the data-parallel listener belong to a TRAINER (FlatAdamW(ctx=...)), not to the process (class attributes, one
    backward pass / one listener / one seed per process).  Two predictors of different shapes with their own optimisers, stepped
    alternately in one process, must end with the parameters - bit for bit - that each reaches when it runs alone; and one
    trainer's range event must not switch the other's weight gradients to the fallback arithmetic."""
    import hashlib
    from npvp_amd import ops
    dev = torch.device(DEV)
    shapes = {"A": (2, 3, 4, 181), "B": (1, 2, 5, 281)}

    def make(tag):
        N, To, Tp, seed = shapes[tag]
        m = GC._small_predictor(impl, False, seed, DEV, evt_layers=1, dec_layers=2, To=To, Tp=Tp, dropout=0.1, drop_path=0.1)
        m.train()
        opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0, ctx=ops.StepContext(tag))
        opt.ctx.rng.manual_seed(seed, dev)
        past = O.synth_features((N, To, 512, 8, 8), seed + 1).to(DEV)
        fut = O.synth_features((N, Tp, 512, 8, 8), seed + 2).to(DEV)
        return m, opt, past, fut

    def digest(opt):
        torch.cuda.synchronize()
        return hashlib.sha256(opt.flat_p.cpu().numpy().tobytes()).hexdigest()

    alone = {}
    for tag in shapes:
        m, opt, past, fut = make(tag)
        for _ in range(3):
            impl.predictor_train_step(m, opt, past, fut, 0.01, 1e-6, 1.0, sync=False)
        alone[tag] = digest(opt)
        del m, opt
    a, b = make("A"), make("B")
    assert a[1].ctx is not b[1].ctx and a[1].ctx is not ops.current()
    for i in range(3):
        impl.predictor_train_step(a[0], a[1], a[2], a[3], 0.01, 1e-6, 1.0, sync=False)
        impl.predictor_train_step(b[0], b[1], b[2], b[3], 0.01, 1e-6, 1.0, sync=False)
    assert digest(a[1]) == alone["A"], "trainer A stepped beside trainer B differs from A alone"
    assert digest(b[1]) == alone["B"], "trainer B stepped beside trainer A differs from B alone"
    assert a[1].ctx.rng.seed is not b[1].ctx.rng.seed
    with ops.use(b[1].ctx):
        ops.RangeGuard.fallback = True                     # B's guard fires (through the module-level name, under B's context) ...
    assert b[1].ctx.range_guard.fallback
    assert not a[1].ctx.range_guard.fallback and not ops.RangeGuard.fallback      # ... A's and the default context's do not
    ops.WgradStream.BATCH = ops.WgradStream.BATCH          # (configuration goes to the class: one knob for every trainer)
    assert "BATCH" not in a[1].ctx.wgrad.__dict__


@pytest.mark.parametrize("variant", ["D", "S"])
def test_activation_gradients_summed_in_place(impl, variant):
    """ops.ActSink / ops.fan_out: the gradients of the positional tables, of the event latent and of the decoder's memory / fused
    key - each the sum over 8 ... 30 consumers - accumulated in place by the consumers' backward kernels against autograd's add
    kernels: same gradients (summation order differs: equal to rounding), ~60 launches fewer per step."""
    from npvp_amd import ops
    N, To, Tp = 2, 3, 4
    past = O.synth_features((N, To, 512, 8, 8), 92).to(DEV)
    fut = O.synth_features((N, Tp, 512, 8, 8), 93).to(DEV)
    old = ops.ActSink.enabled
    res, launches = {}, {}
    try:
        for on in (True, False):
            ops.ActSink.enabled = on
            m = GC._small_predictor(impl, variant == "S", 101, DEV, To=To, Tp=Tp, dropout=0.0, drop_path=0.0)
            m.train()
            opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
            ops.rng.manual_seed(777, torch.device(DEV))
            ops.rng.begin_step(torch.device(DEV))
            opt.zero_grad()
            torch.manual_seed(5)
            p_in = past.clone().requires_grad_(True)
            torch.cuda.synchronize()
            k0 = torch.cuda.memory_stats()["num_alloc_retries"]          # (any cheap counter: the launch count below is what is compared)
            n0 = impl._lib.lib().npvp_launch_count()
            out = m(p_in, fut) if variant == "S" else m(p_in)
            y = out[0] if variant == "S" else out
            (y - fut).abs().mean().backward()
            ops.WgradStream.join()
            torch.cuda.synchronize()
            launches[on] = impl._lib.lib().npvp_launch_count() - n0
            res[on] = (opt.flat_g.clone(), p_in.grad.clone())
        for a, b, what in ((res[True][0], res[False][0], "parameter gradients"), (res[True][1], res[False][1], "input gradient")):
            assert GC.rel_err(a, b) < 1e-6, f"{what}: {GC.rel_err(a, b):.3e}"
        n = res[True][0].numel() // 1024 * 1024
        assert GC.max_row_rel_err(res[True][0][:n].view(-1, 1024), res[False][0][:n].view(-1, 1024), 1e-9) < 1e-4
    finally:
        ops.ActSink.enabled = old


def test_deferred_parameter_gradient_reductions(impl):
    """ops.ReduceQueue: the LayerNorm / frame-LayerNorm / depthwise parameter-gradient partials summed by a few npvp_sum_rows_multi
    launches when the backward pass ends, against one reduction launch per site (same partials, a different split of the partial
    rows over a block's threads: equal to rounding)."""
    from npvp_amd import ops
    N, To, Tp = 2, 3, 4
    past = O.synth_features((N, To, 512, 8, 8), 92).to(DEV)
    fut = O.synth_features((N, Tp, 512, 8, 8), 93).to(DEV)
    old = ops.ReduceQueue.enabled
    flat, launches = {}, {}
    try:
        for on in (True, False):
            ops.ReduceQueue.enabled = on
            m = GC._small_predictor(impl, True, 101, DEV, To=To, Tp=Tp, dropout=0.0, drop_path=0.0)      # NPVP-S: the encoder runs twice
            m.train()
            opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
            ops.rng.manual_seed(777, torch.device(DEV))
            ops.rng.begin_step(torch.device(DEV))
            opt.zero_grad()
            torch.manual_seed(5)
            n0 = impl._lib.lib().npvp_launch_count()
            out = m(past, fut)
            (out[0] - fut).abs().mean().backward()
            ops.WgradStream.join()
            torch.cuda.synchronize()
            launches[on] = impl._lib.lib().npvp_launch_count() - n0
            assert not ops.ReduceQueue.pending()
            flat[on] = opt.flat_g.clone()
        assert launches[True] < launches[False] - 20, launches
        assert GC.rel_err(flat[True], flat[False]) < 1e-6
        n = flat[True].numel() // 1024 * 1024         # chunk by chunk: no gradient slice was skipped or doubled
        assert GC.max_row_rel_err(flat[True][:n].view(-1, 1024), flat[False][:n].view(-1, 1024), 1e-9) < 1e-4
    finally:
        ops.ReduceQueue.enabled = old


def test_stream_experiments_keep_the_amax_slots_ordered(impl):
    """the opt-in stream experiment (NPVP_DUAL_ENCODER: the two NPVP-S encoder passes on two streams, forward and
    backward) cuts fp16 amax slots inside its auxiliary-stream region.  A slot chunk is per (device, stream) - zero-filled on the
    stream that cuts from it - so a step with the switch on must still meet the reference's training-step vectors (a slot that
    read 0 would flush 1e-8-sized gradients)."""
    from npvp_amd import ops
    if MODE != "f16x3":
        pytest.skip("amax slots belong to the f16x3 arithmetic")
    old = ops.AuxStream.enabled
    try:
        ops.AuxStream.enabled = True
        ops.AmaxSlot.reset_chunks()
        mk = lambda m: impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
        res = GC.case_train_step(impl, DEV, "S", make_opt=mk)
        g = GC.load("train_step_S")
        GC.compare({k: v for k, v in res.items() if k.endswith("_0")}, {k: v for k, v in g.items() if k.endswith("_0")}, TOL,
                   tag="train_step_S[dual_encoder]")
        GC.compare({k: v for k, v in res.items() if k.endswith("_1")}, {k: v for k, v in g.items() if k.endswith("_1")}, max(3e-3, TOL))
    finally:
        ops.AuxStream.enabled = old
        ops.AmaxSlot.reset_chunks()
        torch.cuda.synchronize()


def test_grad_sink_matches_autograd_accumulation(impl):
    """ops.GradSink (backward kernels accumulate parameter gradients straight into the flat gradient buffer) against
    the plain autograd route (temporaries + AccumulateGrad adds): same flat gradient, tied LayerNorm and the twice-used
    NPVP-S encoder included."""
    from npvp_amd import ops
    N, To, Tp = 2, 3, 4
    past = O.synth_features((N, To, 512, 8, 8), 92).to(DEV)
    fut = O.synth_features((N, Tp, 512, 8, 8), 93).to(DEV)
    eps = O.seeded_randn((N, 512, 8, 8), 94).to(DEV)
    flat, old = {}, ops.GradSink.enabled
    try:
        for on in (True, False):
            ops.GradSink.enabled = on
            m = GC._small_predictor(impl, True, 101, DEV)
            m.evt_prior.eps_fn = m.evt_posterior.eps_fn = lambda shape: eps
            m.train()
            opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
            opt.zero_grad()
            pred, mu_o, lv_o, mu_p, lv_p = m(past, fut)
            loss = impl.L1Loss(lam=0.01)(pred, fut) + impl.Div_KL(1e-6)(mu_o, lv_o, mu_p, lv_p)
            loss.backward()
            flat[on] = opt.flat_g.clone()
            for p_ in m.parameters():       # every .grad is still the flat slice
                assert p_.grad is not None and p_.grad.data_ptr() >= opt.flat_g.data_ptr()
    finally:
        ops.GradSink.enabled = old
    assert float(flat[True].abs().sum()) > 0
    err = float((flat[True] - flat[False]).norm() / flat[False].norm())
    assert err < 2e-6, f"grad sink vs autograd accumulation: rel-L2 {err:.3e}"


def test_unified_random_context(impl):
    GC.compare(GC.case_randctx(impl, DEV), GC.load("predictor_randctx_S"), TOL, tag=f"randctx[{MODE}]")


def test_reset_pos_coor_fractional_times(impl):
    """continuous-time queries (ref Predictor.py:352-359): fractional context / target times and a new target count on a
    model built for integer steps, against the reference-generated vectors"""
    GC.compare(GC.case_fractime(impl, DEV), GC.load("predictor_fractime_D"), TOL, tag=f"fractime[{MODE}]")


def test_optimizer_state_is_torch_adamw_format(impl, tmp_path):
    """FlatAdamW.state_dict() is what torch.optim.AdamW / a Lightning checkpoint's optimizer_states[0] hold
    (parameter order = predictor.parameters()): it loads into a stock AdamW and round-trips through a .ckpt."""
    m = GC._small_predictor(impl, False, 171, DEV, evt_layers=1, dec_layers=1)
    m.train()
    opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
    past = O.synth_features((1, 3, 512, 8, 8), 172).to(DEV); fut = O.synth_features((1, 4, 512, 8, 8), 173).to(DEV)
    for _ in range(2):
        impl.predictor_train_step(m, opt, past, fut, 0.01, 1e-6, 1.0)
    sd = opt.state_dict(m)
    stock = torch.optim.AdamW(m.parameters(), lr=1e-4)
    stock.load_state_dict(sd)
    p0 = next(m.parameters())
    assert float(stock.state[p0]["step"]) == 2.0 and stock.state[p0]["exp_avg"].shape == p0.shape
    assert float(stock.state[p0]["exp_avg"].abs().sum()) > 0
    path = str(tmp_path / "step2.ckpt")
    impl.save_lightning_checkpoint(path, m, opt=opt, epoch=1, global_step=2, scheduler_T0=150)
    m2 = GC._small_predictor(impl, False, 999, DEV, evt_layers=1, dec_layers=1)
    opt2 = impl.FlatAdamW(m2, lr=3e-4, clip_module=m2.transformer, max_grad_norm=1.0)
    assert impl.load_lightning_checkpoint(path, m2, opt=opt2) == (1, 2)
    assert torch.equal(opt2.flat_p, opt.flat_p) and torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v)
    assert float(opt2.hyper[1]) == 2.0 and opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"]
    # the scheduler record is what torch's CosineAnnealingWarmRestarts holds after `epoch` epoch-steps from the CONSTRUCTION LR
    # (the reference steps it once per epoch): it loads into a real scheduler, which then continues the same cosine
    rec = torch.load(path, weights_only=True)["lr_schedulers"][0]
    stock2 = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(stock2, 150, T_mult=1, eta_min=1e-7)
    sch.step()                                                  # epoch 1, as the record says
    want = sch.state_dict()
    assert rec["_step_count"] == want["_step_count"] and rec["base_lrs"] == want["base_lrs"] == [1e-4]
    assert rec["T_cur"] == want["T_cur"] and rec["last_epoch"] == want["last_epoch"] and rec["T_i"] == want["T_i"]
    sch2 = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1e-4), 150,
                                                               T_mult=1, eta_min=1e-7)
    sch2.load_state_dict(rec)
    sch.step(); sch2.step()
    assert abs(sch.get_last_lr()[0] - sch2.get_last_lr()[0]) < 1e-12
    a = impl.predictor_train_step(m, opt, past, fut, 0.01, 1e-6, 1.0)
    b = impl.predictor_train_step(m2, opt2, past, fut, 0.01, 1e-6, 1.0)
    assert abs(a["loss"] - b["loss"]) <= 1e-6 * abs(a["loss"]) and torch.equal(opt2.flat_p, opt.flat_p)


def test_module_zero_grad_set_to_none_flow(impl):
    """LitPredictor clears gradients with predictor.zero_grad() (ref Predictor.py:126; set_to_none in current torch),
    which detaches .grad from the flat buffer.  FlatAdamW.step() must still see the gradients: same update as the
    trainer's own flow."""
    past = O.synth_features((2, 3, 512, 8, 8), 192).to(DEV); fut = O.synth_features((2, 4, 512, 8, 8), 193).to(DEV)
    res = {}
    for flow in ("flat", "set_to_none"):
        m = GC._small_predictor(impl, False, 191, DEV, evt_layers=1, dec_layers=1)
        m.train()
        opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
        for _ in range(2):
            if flow == "flat":
                opt.zero_grad()
            else:
                m.zero_grad(set_to_none=True)
            loss = impl.L1Loss(lam=0.01)(m(past), fut)
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        res[flow] = opt.flat_p.clone()
        assert all(p_.grad is v_ for p_, v_ in zip(opt.buf.params, opt.buf.views))
    err = float((res["flat"] - res["set_to_none"]).norm() / res["flat"].norm())
    assert err < 1e-6, f"set_to_none flow vs flat flow: rel-L2 {err:.3e}"


def test_two_term_weight_gradients_opt_in(impl):
    """NPVP_WGRAD=bf16x3 (two-term weight-gradient GEMMs, the OPT-IN fast mode) against the reference's training-step
    vectors; the default (six-term weight gradients, same arithmetic as forward / dgrad) is what every other test runs."""
    from npvp_amd import ops
    assert ops.WGRAD_PRECISION is None or os.environ.get("NPVP_WGRAD") == "bf16x3", "six-term weight gradients must be the default"
    if MODE not in ("f16x3", "bf16x6"):
        pytest.skip("the two-term bf16 weight-gradient switch belongs to the bf16x6 mode (run once, from the default mode)")
    old, old_mode = ops.WGRAD_PRECISION, ops.GEMM_PRECISION
    ops.set_gemm_precision("bf16x6")
    ops.WGRAD_PRECISION = 5
    try:
        mk = lambda m: impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
        res = GC.case_train_step(impl, DEV, "S", make_opt=mk)
        g = GC.load("train_step_S")
        GC.compare({k: v for k, v in res.items() if k.endswith("_0")}, {k: v for k, v in g.items() if k.endswith("_0")}, TOL,
                   tag=f"train_step_S_wgrad3[{MODE}]")
        GC.compare({k: v for k, v in res.items() if k.endswith("_1")}, {k: v for k, v in g.items() if k.endswith("_1")}, max(3e-3, TOL))
    finally:
        ops.WGRAD_PRECISION = old
        ops.GEMM_PRECISION = old_mode


def test_training_step_is_bitwise_deterministic(impl):
    """Dropout active, gradient stream on: two identical runs give bit-identical parameters (every in-place gradient
    write is serialised on one stream in program order, every reduction has a fixed order) - a race would show here."""
    import hashlib
    past = O.synth_features((2, 3, 512, 8, 8), 202).to(DEV); fut = O.synth_features((2, 4, 512, 8, 8), 203).to(DEV)
    digests = []
    for _ in range(2):
        m = GC._small_predictor(impl, True, 201, DEV, evt_layers=2, dec_layers=2, dropout=0.1, drop_path=0.1)
        m.train()
        opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
        impl.ops.rng.manual_seed(9, torch.device(DEV))
        torch.manual_seed(3)                      # the NPVP-S reparameterisation noise comes from torch.randn
        for _ in range(3):
            impl.predictor_train_step(m, opt, past, fut, 0.01, 1e-6, 1.0, sync=False)
        torch.cuda.synchronize()
        digests.append(hashlib.sha256(opt.flat_p.cpu().numpy().tobytes()).hexdigest())
    assert digests[0] == digests[1]


@pytest.mark.parametrize("single_stream", [True, False])
def test_graphed_step_equals_eager(impl, single_stream):
    """GraphedTrainStep (the whole optimisation step captured into a HIP graph - on one stream, the default, or with the gradient
    stream inside the capture) replays to the same parameters as the eager step; the dropout seed, lr and step count live in
    device memory."""
    past = O.synth_features((2, 3, 512, 8, 8), 182).to(DEV); fut = O.synth_features((2, 4, 512, 8, 8), 183).to(DEV)
    runs = {}
    for graphed in (False, True):
        m = GC._small_predictor(impl, False, 181, DEV, evt_layers=1, dec_layers=2, dropout=0.1, drop_path=0.1)
        m.train()
        opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
        impl.ops.rng.manual_seed(77, torch.device(DEV))
        if graphed:
            step = impl.GraphedTrainStep(m, opt, past, fut, 0.01, 1e-6, 1.0, warmup=1, single_stream=single_stream)
            assert step.launches > 100
            # the warm-up and the capture executed steps of their own: rewind to the same starting point
            O.key_hashed_fill(m, 181)
            opt.m.zero_(); opt.v.zero_(); opt.hyper[1:2].zero_()
            impl.ops.rng.manual_seed(77, torch.device(DEV))
            for _ in range(3):
                out = step(past, fut, lr=1e-4)
        else:
            for _ in range(3):
                out = impl.predictor_train_step(m, opt, past, fut, 0.01, 1e-6, 1.0, sync=False)
        torch.cuda.synchronize()
        runs[graphed] = (opt.flat_p.clone(), float(out["loss"]))
    (pe, le), (pg, lg) = runs[False], runs[True]
    assert abs(le - lg) <= 1e-5 * abs(le), (le, lg)
    err = float((pe - pg).norm() / pe.norm())
    assert err < 1e-6, f"graph replay vs eager parameters after 3 steps: rel-L2 {err:.3e}"


def test_graphed_step_polls_the_range_guard(impl):
    """A replayed graph cannot be switched to the bf16x6 weight gradients, and nobody reads the range counter inside a replay
    loop: GraphedTrainStep copies the counter to pinned memory every `poll_every` replays without blocking and, on an
    event, captures the step again with RangeGuard.fallback set.  The event is injected here (real training tensors never raise
    it: profiles/r04_f16_range_audit.txt)."""
    if MODE != "f16x3":
        pytest.skip("the range guard belongs to the fp16 arithmetic")
    RG = impl.ops.RangeGuard
    RG.reset()
    past = O.synth_features((2, 3, 512, 8, 8), 182).to(DEV); fut = O.synth_features((2, 4, 512, 8, 8), 183).to(DEV)
    m = GC._small_predictor(impl, False, 181, DEV, evt_layers=1, dec_layers=1, dropout=0.0, drop_path=0.0)
    m.train()
    opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
    try:
        step = impl.GraphedTrainStep(m, opt, past, fut, 0.01, 1e-6, 1.0, warmup=1, poll_every=1)
        step(); step()
        torch.cuda.synchronize()
        assert step.recaptures == 0 and step.range_events == 0
        RG.flag(torch.device(DEV)).fill_(3)              # three weight-gradient launches met an out-of-range feature
        step()                                           # issues the copy of the counter
        torch.cuda.synchronize()
        out = step()                                     # sees it: re-captures with the bf16x6 weight gradients, replays
        torch.cuda.synchronize()
        assert step.recaptures == 1 and step.range_events == 3 and RG.fallback and RG.events == 3
        assert int(RG.flag(torch.device(DEV)).item()) == 0
        assert math.isfinite(float(out["loss"]))
        step(); step()
        torch.cuda.synchronize()
        assert step.recaptures == 1                      # (sticky: nothing further to switch)
    finally:
        RG.reset()


def test_graphed_step_recapture_stays_on_the_eager_trajectory(impl):
    """a re-capture (range event) used to run the constructor's warm-up again - three extra REAL optimiser steps on the
    batch in the static buffers: extra AdamW updates, the step count and the dropout seed advanced behind the caller's back.  A
    re-capture now executes nothing; six calls with a re-capture in the middle leave the parameters, the Adam step count and the loss
    where six eager steps (weight gradients switched to bf16x6 at the same step) leave them."""
    if MODE != "f16x3":
        pytest.skip("the range guard belongs to the fp16 arithmetic")
    RG = impl.ops.RangeGuard
    past = O.synth_features((2, 3, 512, 8, 8), 182).to(DEV); fut = O.synth_features((2, 4, 512, 8, 8), 183).to(DEV)
    runs = {}
    for graphed in (False, True):
        RG.reset()
        m = GC._small_predictor(impl, False, 181, DEV, evt_layers=1, dec_layers=1, dropout=0.1, drop_path=0.1)
        m.train()
        opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
        impl.ops.rng.manual_seed(77, torch.device(DEV))
        try:
            if graphed:
                step = impl.GraphedTrainStep(m, opt, past, fut, 0.01, 1e-6, 1.0, warmup=1, poll_every=1)
                O.key_hashed_fill(m, 181)                   # (rewind: the constructor's warm-up step was a real one)
                opt.m.zero_(); opt.v.zero_(); opt.hyper[1:2].zero_()
                impl.ops.rng.manual_seed(77, torch.device(DEV))
                step(); step()
                torch.cuda.synchronize()
                RG.flag(torch.device(DEV)).fill_(3)
                step()                                      # call 3: issues the copy of the counter (fp16 weight gradients still)
                torch.cuda.synchronize()
                step()                                      # call 4: sees the event, re-captures with bf16x6 weight gradients, replays
                torch.cuda.synchronize()
                assert step.recaptures == 1
                step(); out = step()
                # (a captured step has no memset node: the ROCm 7.2 prepared-packet replay does not execute them reliably, trainer._require_memset_free)
                assert step.census["kernel"] > 100 and step.census.get("memset", 0) == 0, step.census
            else:
                for i in range(6):
                    if i == 3:
                        RG.fallback = True                  # (what the graphed run switches to at its fourth call)
                    out = impl.predictor_train_step(m, opt, past, fut, 0.01, 1e-6, 1.0, sync=False)
            torch.cuda.synchronize()
            runs[graphed] = (opt.flat_p.clone(), float(out["loss"]), float(opt.hyper[1]))
        finally:
            RG.reset()
    (pe, le, ne), (pg, lg, ng) = runs[False], runs[True]
    assert ne == ng == 6.0, (ne, ng)
    assert abs(le - lg) <= 1e-5 * abs(le), (le, lg)
    err = float((pe - pg).norm() / pe.norm())
    assert err < 1e-6, f"six calls with a re-capture in the middle vs six eager steps: parameters rel-L2 {err:.3e}"


def test_captured_steps_have_no_memset_nodes(impl):
    """On ROCm 7.2 a graph replayed from prepared packets (the runtime's default mode) does not execute its memset nodes reliably
    (stale fill patterns or no effect: tools/graph_memset_node_repro.py, profiles/r06_graph_alloc_hazard.txt), so the captured step contains none: zero fills are kernels and the
    loss reductions are the library's fixed-order sums - also with a stochastic predictor (NPVP-S: Div_KL's sum, the
    reparameterisation noise).  `census` is the node count of the capture (npvp_graph_node_counts); the replays in that runtime mode
    are checked by tests/test_dp_gpu.py::test_replayed_step_survives_caller_allocations in processes of their own."""
    if MODE != "f16x3":
        pytest.skip("one arithmetic mode is enough for a property of the captured graph")
    past = O.synth_features((2, 3, 512, 8, 8), 182).to(DEV); fut = O.synth_features((2, 4, 512, 8, 8), 183).to(DEV)
    for stochastic in (False, True):
        m = GC._small_predictor(impl, stochastic, 181, DEV, evt_layers=1, dec_layers=1, dropout=0.1, drop_path=0.1)
        m.train()
        opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
        step = impl.GraphedTrainStep(m, opt, past, fut, 0.01, 1e-6, 1.0, warmup=1)
        assert step.census["kernel"] > 100 and step.census.get("memset", 0) == 0, (stochastic, step.census)
        step()
        torch.cuda.synchronize()


def test_predictor_full_depth(impl):
    GC.compare(GC.case_predictor_full(impl, DEV), GC.load("predictor_full_D"), TOL, tag=f"predictor_full[{MODE}]")


def test_state_dict_roundtrip_with_oracle(impl):
    """Same 603 keys: an oracle (= reference-layout) state dict loads into the HIP Predictor and back."""
    import oracle
    h = torch.linspace(0, 7, 8)
    args = (8, 8, 5, h, h, torch.linspace(0, 1, 2), torch.linspace(2, 4, 3), 512, 'Add', 'layer', 256, 1, True, 1)
    kw = dict(evt_former=True, learn_evt_token=False, evt_former_num_layers=1)
    a, b = oracle.Predictor(*args, **kw), impl.Predictor(*args, **kw)
    O.key_hashed_fill(a, 5)
    b.load_state_dict(a.state_dict())
    a.load_state_dict(b.state_dict())
    assert b.EVT_Former.norm is b.transformer.norm


_LARGER_ORACLE = {}


def _larger_oracle(request, case):
    """the oracle side of one case: from the session's CPU-only worker (tests/larger_oracle.py, started by conftest.py before the GPU
    tests so that it runs beside them) if its file shows up, else computed here; remembered for the other GEMM modes"""
    import time
    import larger_oracle as LO
    if case in _LARGER_ORACLE:
        return _LARGER_ORACLE[case]
    job = getattr(request.config, "_npvp_larger_oracle", None)
    if job is not None:
        proc, outdir = job
        path = os.path.join(outdir, LO.name_of(case) + ".pt")
        t0 = time.time()
        while not os.path.exists(path) and proc.poll() is None and time.time() - t0 < 600:
            time.sleep(0.5)
        if os.path.exists(path):
            d = torch.load(path)
            variant, N, To, Tp, seed0, depth = case
            past, fut = O.synth_features((N, To, 512, 8, 8), d["seed"]), O.synth_features((N, Tp, 512, 8, 8), d["seed"] + 1)
            _LARGER_ORACLE[case] = (d["seed"], past, fut, d["want"])
            return _LARGER_ORACLE[case]
    _LARGER_ORACLE[case] = LO.compute(case)
    return _LARGER_ORACLE[case]


import larger_oracle as _LO


@pytest.mark.parametrize("variant,N,To,Tp,seed0,depth", _LO.CASES)
def test_against_oracle_larger(request, impl, variant, N, To, Tp, seed0, depth):
    """Full depth (4+8), every BASELINE config's clip shape - c0 (S, 5+15), c2' (D, 2+18), c2 (D, 2+28), c3 (S, 2+12),
    c4 (D, 4+16), c1 (S, 10+10) - and (round 4) the WHOLE per-GPU shard of the 8-GPU configuration c4 (8 clips of 4 + 16:
    8 192 decoder token rows, i.e. the shapes and kernel variants the data-parallel benchmark line runs; at depth 1 + 2 - every
    layer has the same shapes, and at full depth the CPU oracle needed 4.5 minutes for this one case; default arithmetic only: the
    exact-fp32 mode runs the same shapes in the cases above); round 5: a clip of 3 + 40
    frames - longer than any shipped configuration, the temporal (40 x 40) and encoder-decoder (40 x 3) attention on the generic
    kernels: HIP vs oracle on the same seeded inputs, forward (train mode, dropout 0) and gradients.  The oracle side (20 - 40 s of
    CPU per case) comes from a CPU-only worker that runs beside the GPU tests (tests/larger_oracle.py)."""
    case = (variant, N, To, Tp, seed0, depth)
    if N == 8 and MODE != DEFAULT_MODE:
        pytest.skip("the 8-clip shard case runs in the default arithmetic only (VERDICT r5 item 8)")
    seed, past, fut, want = _larger_oracle(request, case)
    args, kw = _LO.predictor_args(case)
    hip = impl.Predictor(*args, **kw)
    O.key_hashed_fill(hip, 7)
    got = _LO.run(hip.to(DEV), DEV, past, fut, case)
    outs = [want, got]
    for a, b, n in zip(outs[1], outs[0], ["y", "g_past", "g_tied_norm"]):
        e = GC.rel_err(a, b)
        GC.log_err(f"larger_{variant}[{MODE}]", n, e)
        assert e < TOL * 5, f"{n}: {e:.3e} (input seed {seed})"


@pytest.mark.parametrize("name,variant,B,To,Tp", [("c2", "D", 64, 2, 28), ("c2p", "D", 64, 2, 18), ("c3 shard", "S", 8, 2, 12),
                                                  ("c4 shard", "D", 8, 4, 16)])
def test_full_size_properties_other_configs(impl, name, variant, B, To, Tp):
    """BASELINE c2 (BAIR NPVP-D, B=64, 2+28), north_star's B=64 T=20 line and the per-GPU shards of c3 / c4 at FULL size,
    dropout 0.1 / drop-path 0.1 active: training steps run, losses finite and decreasing on a fixed batch, gradients
    finite, eval output deterministic, non-negative and of the right shape."""
    if MODE != DEFAULT_MODE:
        pytest.skip("full-size runs use the default arithmetic")
    torch.manual_seed(0)
    stochastic = variant == "S"
    h = torch.linspace(0, 7, 8)
    m = impl.Predictor(8, 8, To + Tp, h, h, torch.linspace(0, To - 1, To), torch.linspace(To, To + Tp - 1, Tp), 512, 'Add', 'layer',
                       256, 1, stochastic, 8, evt_former=True, learn_evt_token=False, evt_former_num_layers=4).to(DEV)
    opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer)
    past, fut = O.synth_features((B, To, 512, 8, 8), 1).to(DEV), O.synth_features((B, Tp, 512, 8, 8), 2).to(DEV)
    m.train()
    losses = [impl.predictor_train_step(m, opt, past, fut, 0.01, 1e-8, 1.0)["loss"] for _ in range(3)]
    assert all(l == l and abs(l) < 1e3 for l in losses), losses
    assert losses[-1] < losses[0], losses
    assert bool(torch.isfinite(opt.flat_g).all())
    m.eval()
    if stochastic:
        eps = O.seeded_randn((B, 512, 8, 8), 9).to(DEV)
        m.evt_prior.eps_fn = lambda shape: eps
    with torch.no_grad():
        y1, y2 = m(past), m(past)
    assert torch.equal(y1, y2), "eval forward must be bit-deterministic"
    assert bool((y1 >= 0).all()) and y1.shape == (B, Tp, 512, 8, 8)
    del m, opt, past, fut, y1, y2
    torch.cuda.empty_cache()


def test_full_size_properties(impl):
    """BASELINE c1 size (KTH NPVP-S, B=32, To=Tp=10, dropout 0.1 / drop-path 0.1 active): one training step runs,
    output is finite, non-negative (final ReLU), gradients finite, loss decreases over a few steps on a fixed batch,
    and eval mode is deterministic."""
    torch.manual_seed(0)
    h = torch.linspace(0, 7, 8)
    m = impl.Predictor(8, 8, 20, h, h, torch.linspace(0, 9, 10), torch.linspace(10, 19, 10), 512, 'Add', 'layer', 256, 1,
                       True, 8, evt_former=True, learn_evt_token=False, evt_former_num_layers=4).to(DEV)
    opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer)
    past, fut = O.synth_features((32, 10, 512, 8, 8), 1).to(DEV), O.synth_features((32, 10, 512, 8, 8), 2).to(DEV)
    m.train()
    losses = [impl.predictor_train_step(m, opt, past, fut, 0.01, 1e-8, 1.0)["loss"] for _ in range(4)]
    assert all(l == l and abs(l) < 1e3 for l in losses), losses
    assert losses[-1] < losses[0], losses
    assert bool(torch.isfinite(opt.flat_g).all())
    m.eval()
    eps = O.seeded_randn((32, 512, 8, 8), 9).to(DEV)
    m.evt_prior.eps_fn = lambda shape: eps        # NPVP-S samples z from the prior in eval too (ref Predictor.py:321)
    with torch.no_grad():
        y1, y2 = m(past), m(past)
    assert torch.equal(y1, y2), "eval forward must be bit-deterministic"
    assert bool((y1 >= 0).all()) and y1.shape == (32, 10, 512, 8, 8)


@pytest.mark.parametrize("path", ["stock", "fused"])
@pytest.mark.parametrize("tag", ["64", "128"])
def test_frozen_decoder_input_gradient_with_shared_relu_masks(impl, tag, path):
    """The TIGHT form of test_frozen_autoencoder's g_feats check.  d(frames)/d(feats) passes through the decoder's ReLU masks,
    and a unit within rounding noise of 0 lands on either side depending on the convolution algorithm (millions of units: no
    input avoids them all), which is why the plain comparison above is held to 5e-3.  Here the HIP decoder's OWN masks are read
    back and injected into an fp64 CPU twin of the decoder (same weights, ReLU replaced by 'multiply by this mask'): with the
    kink decisions shared, the input gradient must agree to 1e-4 - transposed convolutions, folded BatchNorm, csrc/ae.hip's
    act_bwd and all - and the two mask sets may differ in at most 1e-4 of the units."""
    import torch.nn as nn
    if MODE != DEFAULT_MODE:
        pytest.skip("the frozen autoencoder has no GEMM of the predictor's kind: one arithmetic mode is enough")
    ci, ngf, nd, nres, S, out_layer = {"64": (1, 64, 3, 2, 64, 'Sigmoid'), "128": (3, 32, 4, 3, 128, 'Tanh')}[tag]
    enc = impl.ResnetEncoder(ci, ngf=ngf, n_downsampling=nd, num_res_blocks=nres, learn_3d=False)
    dec = impl.ResnetDecoder(ci, ngf=ngf, n_downsampling=nd, out_layer=out_layer)
    twin = impl.ResnetDecoder(ci, ngf=ngf, n_downsampling=nd, out_layer=out_layer)
    O.key_hashed_fill(enc, 121); O.key_hashed_fill(dec, 122); O.key_hashed_fill(twin, 122)
    enc, dec, twin = enc.eval(), dec.eval(), twin.double().eval()
    if path == "stock":
        enc, dec = enc.to(DEV), dec.to(DEV)
    else:
        for q in list(enc.parameters()) + list(dec.parameters()):
            q.requires_grad_(False)
        enc, dec = impl.to_device_layout(enc, dec, DEV)
    relu_sites = [m for m in dec.model if isinstance(m, nn.ReLU) or getattr(m, "act", 0) == 1]
    assert len(relu_sites) == nd
    masks = []
    hooks = [m.register_forward_hook(lambda mod, inp, out: masks.append((out.detach() > 0).cpu())) for m in relu_sites]
    x = torch.rand(1, 2, ci, S, S, generator=torch.Generator().manual_seed(123)).to(DEV)
    with torch.no_grad():
        feats = enc(x)
    f = feats.clone().requires_grad_()
    y = dec(f)
    cot = O.seeded_randn(y.shape, 124)
    (y * cot.to(DEV)).sum().backward()
    for h in hooks:
        h.remove()
    assert len(masks) == nd

    class Masked(nn.Module):
        def __init__(self, mask):
            super().__init__()
            self.mask, self.natural = mask, None

        def forward(self, v):
            self.natural = v.detach() > 0
            return v * self.mask.to(v.dtype)

    it = iter(masks)
    twin.model = nn.Sequential(*[Masked(next(it)) if isinstance(m, nn.ReLU) else m for m in twin.model])
    f64 = feats.detach().double().cpu().requires_grad_()
    (twin(f64) * cot.double()).sum().backward()
    e = GC.rel_err(f.grad.cpu(), f64.grad.float())
    GC.log_err(f"ae_{tag}.g_feats_shared_masks[{MODE},{path}]", "g_feats", e)
    assert e < 1e-4, f"decoder input gradient with shared ReLU masks: rel-L2 {e:.3e}"
    flips = sum(int((m.natural != m.mask).sum()) for m in twin.model if isinstance(m, Masked))
    units = sum(m.mask.numel() for m in twin.model if isinstance(m, Masked))
    assert flips <= 1e-4 * units, f"{flips} of {units} ReLU units on different sides of the kink"


def test_decoder_block_batch8_against_oracle(impl):
    """One full c2 decoder block (T2 = 28 target steps, T1 = 2 context steps) at N = 8 clips against the CPU oracle, forward and
    every input gradient: the BATCH dimension of the benchmarked shape value-checked (the full-depth comparisons above stop at
    N = 1 - 2, the full-size runs check properties only).  14 336 token rows: the fp16 GEMM kernels, the staged T = 28
    attention backward, the fused MlpDWBN middle."""
    import oracle
    N, T2, T1 = 8, 28, 2

    def run(mod_impl, dev):
        m = mod_impl.VidHRFormerBlockDecNAR(8, 8, 512, 8, 4, 0.0, 0.0, 4, 1024)
        O.key_hashed_fill(m, 61)
        m = m.to(dev).train()
        tgt = (0.3 * O.seeded_randn((N, T2, 8, 8, 512), 62)).to(dev).requires_grad_()
        qe = (0.5 * O.seeded_randn((N, 8, 8, 512), 63)).to(dev).requires_grad_()
        mem = O.synth_features((N, T1, 8, 8, 512), 64).to(dev).requires_grad_()
        mb, _ = GC.pos_tables(T1, 65, False, dev)
        tb, _ = GC.pos_tables(T2, 66, False, dev)
        cot = O.seeded_randn((N, T2, 8, 8, 512), 67).to(dev)
        y = m(tgt, qe, mem, (mb, None), (tb, None), mod_impl.PosFeatFuser(512, 'layer'))
        g = torch.autograd.grad((y * cot).sum(), [tgt, qe, mem, m.EncDecAttn.in_proj_weight, m.SpatialFFN.fc1.weight, m.norm5.bias])
        return [t.detach().cpu() for t in (y, *g)]

    key = "dec_block_n8"
    if key not in _LARGER_ORACLE:
        _LARGER_ORACLE[key] = run(oracle, "cpu")
    want, got = _LARGER_ORACLE[key], run(impl, DEV)
    for a, b, n in zip(got, want, ["y", "g_tgt", "g_qe", "g_mem", "g_encdec_W", "g_fc1_W", "g_norm5_b"]):
        e = GC.rel_err(a, b)
        GC.log_err(f"dec_block_n8[{MODE}]", n, e)
        assert e < TOL, f"{n}: {e:.3e}"
