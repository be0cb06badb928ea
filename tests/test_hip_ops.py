"""GPU parity tests, one HIP kernel family at a time, against the CPU oracle (oracle/ops.py) on
the same seeded inputs - forward values and every gradient.  All calls go through the C ABI of
libnpvp_hip.so (npvp_amd.ops -> ctypes).  Tolerances: the fp32-MFMA path is expected to agree to
~1e-6; the bar written here (1e-4 rel-L2, 1e-3 for long reductions) sits inside the 1e-3 rel fp32
bar of BASELINE.json's north_star."""
import math
import os

import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = 1e-4


def rel(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


ROW_TOL = 1e-4       # per ROW (token row of an activation / gradient, output-feature row of a weight gradient); north_star's bar is
                     # 1e-3, measured worst over every op test and fp32-grade mode: 7e-6 (bf16x3: 1.1e-5)


def close(a, b, tol=TOL, what="", row_tol=None, row_floor=1e-6):
    """whole-tensor rel-L2 below `tol` AND the worst row's rel-L2 below `row_tol` (default: max(ROW_TOL, tol))"""
    from golden_cases import max_row_rel_err
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    e = rel(a, b)
    er = max_row_rel_err(a, b, row_floor)
    if os.environ.get("NPVP_ERR_LOG"):
        import npvp_amd.ops as _o
        with open(os.environ["NPVP_ERR_LOG"], "a") as f:
            f.write(f"ops[{_o.GEMM_PRECISION}] {os.environ.get('PYTEST_CURRENT_TEST', '').split('::')[-1].split(' ')[0]} {what} {e:.3e} row {er:.3e}\n")
    assert e < tol, f"{what}: rel-L2 {e:.3e} >= {tol:.1e}"
    rt = max(ROW_TOL, tol) if row_tol is None else row_tol
    assert er < rt, f"{what}: worst-row rel-L2 {er:.3e} >= {rt:.1e}"


# the exact fp32-MFMA kernel, the default three-term split (wide + 128x128 kernels) and its two-term sibling (opt-in wgrad mode)
MODES = ["f32", "f16x3", "bf16x6", "bf16x3"]


@pytest.fixture(scope="module", params=MODES)
def K(request):
    """Every test runs on each GEMM arithmetic path."""
    import npvp_amd
    from npvp_amd import ops
    assert torch.cuda.is_available()
    ops.rng.manual_seed(1234, torch.device(DEV))
    ops.set_gemm_precision(request.param)
    yield ops
    ops.set_gemm_precision("f16x3")


def g(t):
    """leaf copy on the device"""
    return t.detach().to(DEV).requires_grad_(t.requires_grad)


# ------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K_", [(128, 128, 32), (256, 512, 512), (160, 96, 64), (20, 36, 32), (1024, 2048, 512),
                                    (640, 512, 2048), (2080, 224, 96)])
def test_gemm_forward_variants(K, M, N, K_):
    x = O.seeded_randn((M, K_), 1); w = O.seeded_randn((N, K_), 2) / math.sqrt(K_); b = O.seeded_randn((N,), 3)
    r = O.seeded_randn((M, N), 4)
    xg, wg, bg, rg = x.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV)
    close(K.linear_fwd(xg, wg, None), x @ w.T, what="plain")
    close(K.linear_fwd(xg, wg, bg), x @ w.T + b, what="bias")
    aux = torch.empty(M, N, device=DEV)
    y = K.linear_fwd(xg, wg, bg, act=1, aux_out=aux)
    close(aux, x @ w.T + b, what="aux"); close(y, O.gelu(x @ w.T + b), what="gelu")
    close(K.linear_fwd(xg, wg, bg, act=2, residual=rg), torch.relu(x @ w.T + b) + r, what="relu+res")
    # dgrad: dx = dy @ w ; with GELU' epilogue (the reduction dim N must be a multiple of 32)
    dy = O.seeded_randn((M, N), 5)
    if N % 32 == 0:
        close(K.linear_dgrad(dy.to(DEV), wg), dy @ w, what="dgrad")
        hpre = O.seeded_randn((M, K_), 6)
        hp = hpre.clone().requires_grad_()
        (O.gelu(hp) * (dy @ w)).sum().backward()
        close(K.linear_dgrad(dy.to(DEV), wg, act=3, aux_in=hpre.to(DEV)), hp.grad, what="dgrad*gelu'")
    # wgrad: dw = dy^T x  (the reduction dim = token rows, a multiple of 64 on the path)
    if M % 32 == 0:
        dw, db = K.linear_wgrad(dy.to(DEV), xg, True)
        close(dw, dy.T @ x, what="wgrad"); close(db, dy.sum(0), what="fused bias grad")


def test_gemm_wgrad_splitk_long_reduction(K):
    R, N, K_ = 8192, 512, 256
    dy = O.seeded_randn((R, N), 7); x = O.seeded_randn((R, K_), 8)
    from npvp_amd._lib import lib
    assert lib().npvp_gemm_workspace_bytes(N, K_, R) > 0, "expected the split-K path for this shape"
    dw, db = K.linear_wgrad(dy.to(DEV), x.to(DEV), True)
    close(dw, dy.T @ x, tol=5e-5 if K.GEMM_PRECISION in (1, 5) else 1e-5, what="split-K wgrad")
    close(db, dy.sum(0), tol=1e-5, what="split-K fused bias grad")
    close(K.colsum(dy.to(DEV)), dy.sum(0), tol=1e-5, what="colsum")


def test_gemm_full_size_against_rocblas(K):
    """BASELINE c1 sizes (R = 32*10*64 = 20480 rows): a fast independent check of the big GEMM shapes."""
    R = 20480
    for N, K_ in ((2048, 512), (512, 2048), (1024, 512), (512, 512)):
        x = torch.randn(R, K_, device=DEV); w = torch.randn(N, K_, device=DEV) / math.sqrt(K_)
        y = K.linear_fwd(x, w, None)
        ref = (x.double() @ w.double().T).float()
        tol = 5e-5 if K.GEMM_PRECISION in (1, 5) else 1e-5
        close(y, ref, tol=tol, what=f"fwd {N}x{K_}")
        dy = torch.randn(R, N, device=DEV)
        close(K.linear_wgrad(dy, x), (dy.double().T @ x.double()).float(), tol=tol, what=f"wgrad {N}x{K_}")
        close(K.linear_dgrad(dy, w), (dy.double() @ w.double()).float(), tol=tol, what=f"dgrad {N}x{K_}")
    # linearity (size-independent property): f(a x1 + x2) = a f(x1) + f(x2)
    x1, x2 = torch.randn(R, 512, device=DEV), torch.randn(R, 512, device=DEV)
    w = torch.randn(512, 512, device=DEV) / 22.6
    close(K.linear_fwd(2.5 * x1 + x2, w, None), 2.5 * K.linear_fwd(x1, w, None) + K.linear_fwd(x2, w, None), tol=5e-5)


def test_linear_emits_frame_statistics(K):
    """linear(frame_stats=True): the forward GEMM's epilogue yields the frame-LayerNorm statistics of its output
    (frames of 64 token rows) - what MlpDWBN's norm1 / norm3 consume instead of a statistics pass."""
    from npvp_amd import ops
    if not ops.linear_frame_stats_supported(320, 2048):
        pytest.skip("row statistics ride on the default (bf16x6) forward kernels")
    for R, N, K_ in ((5 * 64, 2048, 512), (3 * 64, 512, 2048), (64, 128, 64)):      # odd frame counts: half-empty last tile
        x = (O.seeded_randn((R, K_), 47) + 0.5).to(DEV).requires_grad_()
        w = (O.seeded_randn((N, K_), 48) / math.sqrt(K_)).to(DEV).requires_grad_(); b = (O.seeded_randn((N,), 49) + 2.0).to(DEV).requires_grad_()
        cot = O.seeded_randn((R, N), 50).to(DEV)
        y0 = ops.linear(x, w, b)
        y1, mean, rstd = ops.linear(x, w, b, frame_stats=True)
        close(y1, y0, tol=1e-6, what="output")
        fr = y1.detach().double().reshape(R // 64, -1)
        close(mean, fr.mean(1).float(), tol=1e-6, what="mean")
        close(rstd, (1.0 / torch.sqrt(fr.var(1, unbiased=False) + 1e-5)).float(), tol=2e-6, what="rstd")
        g0 = torch.autograd.grad((y0 * cot).sum(), [x, w, b]); g1 = torch.autograd.grad((y1 * cot).sum(), [x, w, b])
        for a_, b_ in zip(g0, g1):
            assert torch.equal(a_, b_)


def test_linear_autograd(K):
    x = O.seeded_randn((3, 64, 512), 11).requires_grad_(); w = (O.seeded_randn((256, 512), 12) / 22.0).requires_grad_()
    b = O.seeded_randn((256,), 13).requires_grad_(); r = O.seeded_randn((3, 64, 256), 14).requires_grad_()
    cot = O.seeded_randn((3, 64, 256), 15)
    y = torch.nn.functional.linear(x, w, b) + r
    gx, gw, gb, gr = torch.autograd.grad((y * cot).sum(), [x, w, b, r])
    xg, wg, bg, rg = g(x.detach().requires_grad_()), g(w.detach().requires_grad_()), g(b.detach().requires_grad_()), g(r.detach().requires_grad_())
    yg = K.linear(xg, wg, bg, residual=rg)
    hx, hw, hb, hr = torch.autograd.grad((yg * cot.to(DEV)).sum(), [xg, wg, bg, rg])
    close(yg, y); close(hx, gx); close(hw, gw); close(hb, gb); close(hr, gr)


def test_ffn(K):
    R, C, Fh = 384, 512, 1024
    xn = O.seeded_randn((R, C), 21).requires_grad_(); x = O.seeded_randn((R, C), 22).requires_grad_()
    w1 = (O.seeded_randn((Fh, C), 23) / 22).requires_grad_(); b1 = (0.1 * O.seeded_randn((Fh,), 24)).requires_grad_()
    w2 = (O.seeded_randn((C, Fh), 25) / 32).requires_grad_(); b2 = (0.1 * O.seeded_randn((C,), 26)).requires_grad_()
    cot = O.seeded_randn((R, C), 27)
    y = x + O.linear(O.gelu(O.linear(xn, w1, b1)), w2, b2)
    ref = torch.autograd.grad((y * cot).sum(), [xn, x, w1, b1, w2, b2])
    ins = [g(t.detach().requires_grad_()) for t in (xn, x, w1, b1, w2, b2)]
    yg = K.ffn(*ins, 0.0)
    got = torch.autograd.grad((yg * cot.to(DEV)).sum(), ins)
    close(yg, y)
    for a, b_, n in zip(got, ref, ["dxn", "dx", "dw1", "db1", "dw2", "db2"]):
        close(a, b_, what=n)


# ------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("rows,relu", [(64, False), (1000, True), (4099, False)])
def test_layernorm(K, rows, relu):
    x = O.seeded_randn((rows, 512), 31).requires_grad_()
    w = (1 + 0.1 * O.seeded_randn((512,), 32)).requires_grad_(); b = (0.1 * O.seeded_randn((512,), 33)).requires_grad_()
    cot = O.seeded_randn((rows, 512), 34)
    y = O.layernorm(x, w, b)
    y = torch.relu(y) if relu else y
    ref = torch.autograd.grad((y * cot).sum(), [x, w, b])
    ins = [g(t.detach().requires_grad_()) for t in (x, w, b)]
    yg = K.layernorm(ins[0], ins[1], ins[2], 1e-5, relu)
    got = torch.autograd.grad((yg * cot.to(DEV)).sum(), ins)
    close(yg, y)
    for a, b_, n in zip(got, ref, ["dx", "dw", "db"]):
        close(a, b_, what=n)


@pytest.mark.parametrize("with_add,with_gamma,N,T", [(False, False, 3, 4), (True, False, 3, 4), (True, True, 3, 4), (True, True, 18, 2)])
def test_posfuse(K, with_add, with_gamma, N, T):
    """N = 3: the batch sums d beta / d gamma come out of the apply pass (batch loop in the thread); N = 18, T = 2 (more than 16
    samples, few (t, e) columns - the shape of c2's encoder): the dy * uhat scratch and the two reductions inside the library"""
    P, C = 64, 512
    import npvp_amd
    assert npvp_amd.ops.lib().npvp_posfuse_bwd_fused(N, T, P * C) == (1 if N == 3 else 0)
    x = (0.5 + O.seeded_randn((N * T, P, C), 41)).requires_grad_()
    add = O.seeded_randn((N, P, C), 42).requires_grad_() if with_add else None
    beta = O.seeded_randn((T * P, C), 43).requires_grad_()
    gamma = (0.3 * O.seeded_randn((T * P, C), 44)).requires_grad_() if with_gamma else None
    cot = O.seeded_randn((N * T, P, C), 45)
    y = O.posfuse(x, T, beta, gamma, add)
    leaves = [t for t in (x, add, beta, gamma) if t is not None]
    ref = torch.autograd.grad((y * cot).sum(), leaves)
    gl = {id(t): g(t.detach().requires_grad_()) for t in leaves}
    get = lambda t: None if t is None else gl[id(t)]
    yg = K.posfuse(get(x), get(add), get(beta), get(gamma), N, T)
    got = torch.autograd.grad((yg * cot.to(DEV)).sum(), [gl[id(t)] for t in leaves])
    close(yg, y)
    for a, b_ in zip(got, ref):
        close(a, b_)


@pytest.mark.parametrize("Ch,with_res", [(2048, False), (512, True)])
def test_frameln_act(K, Ch, with_res):
    F_, P = 5, 64
    h = (0.3 + O.seeded_randn((F_, P, Ch), 51)).requires_grad_()
    w = (1 + 0.1 * O.seeded_randn((P, Ch), 52)).requires_grad_(); b = (0.1 * O.seeded_randn((P, Ch), 53)).requires_grad_()
    res = O.seeded_randn((F_, P, Ch), 54).requires_grad_() if with_res else None
    cot = O.seeded_randn((F_, P, Ch), 55)
    y = O.gelu(O.frame_ln(h, w, b))
    y = y + res if with_res else y
    leaves = [t for t in (h, w, b, res) if t is not None]
    ref = torch.autograd.grad((y * cot).sum(), leaves)
    gl = [g(t.detach().requires_grad_()) for t in leaves]
    yg = K.frameln_act(gl[0], gl[1].reshape(-1), gl[2].reshape(-1), gl[3] if with_res else None, F_)
    got = torch.autograd.grad((yg * cot.to(DEV)).sum(), gl)
    close(yg, y)
    for a, b_, n in zip(got, ref, ["dh", "dw", "db", "dres"]):
        close(a.reshape(b_.shape), b_, what=n)


def test_dwconv(K):
    F_, H, W, Ch = 6, 8, 8, 256
    a = O.seeded_randn((F_, H * W, Ch), 61).requires_grad_()
    w = (0.3 * O.seeded_randn((Ch, 3, 3), 62)).requires_grad_(); b = (0.1 * O.seeded_randn((Ch,), 63)).requires_grad_()
    cot = O.seeded_randn((F_, H * W, Ch), 64)
    y = O.dwconv3x3(a, w, b, H, W)
    ra, rw, rb = torch.autograd.grad((y * cot).sum(), [a, w, b])
    ag = g(a.detach().requires_grad_())
    wtb = torch.cat([w.detach().reshape(Ch, 9).t(), b.detach().reshape(1, Ch)], 0).contiguous().to(DEV).requires_grad_()
    yg = K.dwconv3x3(ag, wtb, F_, H, W)
    ga, gwtb = torch.autograd.grad((yg * cot.to(DEV)).sum(), [ag, wtb])
    close(yg, y); close(ga, ra)
    close(gwtb[:9].t().reshape(Ch, 3, 3), rw, what="dw"); close(gwtb[9], rb, what="db")


def test_dwconv_emits_frame_statistics(K):
    """dwconv3x3(want_stats=True): same output and gradients, plus the frame-LayerNorm statistics of the output (what
    MlpDWBN's norm2 consumes instead of a statistics pass), also for frames with a large common offset."""
    F_, H, W, Ch = 5, 8, 8, 2048
    a = (O.seeded_randn((F_, H * W, Ch), 65) + torch.arange(F_).view(F_, 1, 1) * 3.0).to(DEV).requires_grad_()
    wtb = torch.cat([0.3 * O.seeded_randn((9, Ch), 66), 0.1 * O.seeded_randn((1, Ch), 67) + 5.0], 0).to(DEV).requires_grad_()
    cot = O.seeded_randn((F_, H * W, Ch), 68).to(DEV)
    y0 = K.dwconv3x3(a, wtb, F_, H, W)
    y1, mean, rstd = K.dwconv3x3(a, wtb, F_, H, W, want_stats=True)
    close(y1, y0, tol=1e-6, what="output")       # (two template instantiations: fma contraction may differ in the last bit)
    flat = y1.detach().double().reshape(F_, -1)
    close(mean, flat.mean(1).float(), tol=1e-6, what="mean")
    close(rstd, (1.0 / torch.sqrt(flat.var(1, unbiased=False) + 1e-5)).float(), tol=1e-6, what="rstd")
    g0 = torch.autograd.grad((y0 * cot).sum(), [a, wtb]); g1 = torch.autograd.grad((y1 * cot).sum(), [a, wtb])
    assert torch.equal(g0[0], g1[0]) and torch.equal(g0[1], g1[1])


# ------------------------------------------------------------------------------- attention cores
def _attn_ref(q, k, v, rows_q, rows_k, mask):
    return O.attn_core(q, k, v, rows_q, rows_k, 8, mask)


@pytest.mark.parametrize("frames,ws", [(1, 4), (3, 4), (2, 8)])
def test_attn_spatial(K, frames, ws):
    """ws = 8: one 64-token window per frame (sequence length 64: the generic kernels)"""
    from npvp_amd.ops import AttnCfg
    R, C = frames * 64, 512
    qk = O.seeded_randn((R, 2 * C), 71).requires_grad_(); v = O.seeded_randn((R, C), 72).requires_grad_()
    cot = O.seeded_randn((R, C), 73)
    rows = O.spatial_groups(frames, 8, 8, ws)
    y = _attn_ref(qk[:, :C], qk[:, C:], v, rows, rows, None)
    rqk, rv = torch.autograd.grad((y * cot).sum(), [qk, v])
    qkg, vg = g(qk.detach().requires_grad_()), g(v.detach().requires_grad_())
    yg = K.attn_packed(qkg, vg, AttnCfg(0, frames, 64, 8, ws, 0, 0, 8, 0, 0.0))
    gqk, gv = torch.autograd.grad((yg * cot.to(DEV)).sum(), [qkg, vg])
    close(yg, y); close(gqk, rqk, what="dqk"); close(gv, rv, what="dv")


@pytest.mark.parametrize("Tq,Tk,mask", [(1, 1, 0), (2, 2, 1), (3, 3, 1), (10, 10, 1), (10, 10, 0), (17, 17, 1), (28, 28, 0),
                                        (32, 32, 1), (4, 2, 0), (18, 2, 0), (10, 28, 0), (28, 10, 0),
                                        # the BASELINE configs' decoder / cross shapes: c2 (28,28),(28,2); c3 (12,12),(12,2); c4 (16,16),(16,4)
                                        (28, 2, 0), (12, 12, 0), (12, 2, 0), (16, 16, 0), (16, 4, 0), (4, 4, 1), (18, 18, 0),
                                        # longer than any shipped configuration (the reference has no limit): the generic kernels
                                        (40, 40, 1), (33, 33, 0), (48, 5, 0), (5, 40, 0), (128, 128, 1)])
def test_attn_temporal_and_cross(K, Tq, Tk, mask):
    from npvp_amd.ops import AttnCfg
    N, P, C = 2, 64, 512
    q = O.seeded_randn((N * Tq * P, C), 81).requires_grad_()
    k = O.seeded_randn((N * Tk * P, C), 82).requires_grad_(); v = O.seeded_randn((N * Tk * P, C), 83).requires_grad_()
    cot = O.seeded_randn((N * Tq * P, C), 84)
    m = O.encoder_temporal_mask(Tq) if mask else None
    y = _attn_ref(q, k, v, O.temporal_groups(N, Tq, P), O.temporal_groups(N, Tk, P), m)
    ref = torch.autograd.grad((y * cot).sum(), [q, k, v])
    ins = [g(t.detach().requires_grad_()) for t in (q, k, v)]
    yg = K.attn(ins[0], ins[1], ins[2], AttnCfg(1, N, P, 8, 0, Tq, Tk, 8, mask, 0.0))
    got = torch.autograd.grad((yg * cot.to(DEV)).sum(), ins)
    close(yg, y)
    for a, b_, n in zip(got, ref, ["dq", "dk", "dv"]):
        close(a, b_, what=n)


# ------------------------------------------------------------------------------- layout / reductions
def test_transpose_and_mean(K):
    x = O.seeded_randn((3, 4, 512, 8, 8), 91)
    xc = K.nchw_to_canonical(x.to(DEV))
    assert torch.equal(xc.cpu(), O.to_canonical(x))
    assert torch.equal(K.canonical_to_nchw(xc, 3, 4, 8, 8).cpu(), x)
    y = O.seeded_randn((3, 5, 1024), 92).requires_grad_()
    yg = g(y.detach().requires_grad_())
    m = K.mean_mid(yg)
    close(m, y.mean(1), tol=1e-6)
    (m * 2).sum().backward(); (y.mean(1) * 2).sum().backward()
    close(yg.grad, y.grad, tol=1e-6)


@pytest.mark.parametrize("shape", [(2, 3, 512, 8, 8), (1, 1, 37), (64, 28, 512, 8, 8)])
def test_scalar_losses_against_the_oracle(shape):
    """L1Loss and Div_KL (oracle/model.py, ref/models/criterion.py:99-121,341-354) through the library's fixed-order sums
    (npvp_l1_mean / npvp_l1_mean_bwd / npvp_sum_all): values to 1e-6, the L1 gradient EQUAL to the one torch's abs -> mean -> mul
    chain produces (sign * lam / n, same order of operations), two runs equal to the bit; a ragged length exercises the tails."""
    import npvp_amd
    from oracle import model as OM
    a, b = O.seeded_randn(shape, 701), O.seeded_randn(shape, 702)
    b.view(-1)[::7] = a.view(-1)[::7]                       # exact zeros of a - b: sgn(0) = 0
    ao = a.clone().requires_grad_()
    want = OM.L1Loss(lam=0.01)(ao, b)
    (want * 3.0).backward()
    ad = a.to(DEV).requires_grad_()
    got = npvp_amd.L1Loss(lam=0.01)(ad, b.to(DEV))
    (got * 3.0).backward()
    assert abs(float(got) - float(want)) <= 1e-6 * abs(float(want))
    assert torch.equal(ad.grad.cpu(), ao.grad)
    again = npvp_amd.L1Loss(lam=0.01)(ad.detach(), b.to(DEV))
    assert torch.equal(again, got.detach())
    if len(shape) == 5 and shape[0] <= 2:
        mu1, lv1, mu2, lv2 = (O.seeded_randn(shape, 710 + i) * 0.3 for i in range(4))
        lo = [t.clone().requires_grad_() for t in (mu1, lv1, mu2, lv2)]
        ld = [t.to(DEV).requires_grad_() for t in (mu1, lv1, mu2, lv2)]
        kw, kg = OM.Div_KL(1e-6)(*lo), npvp_amd.Div_KL(1e-6)(*ld)
        kw.backward(); kg.backward()
        assert abs(float(kg) - float(kw)) <= 2e-6 * abs(float(kw))
        for x, y in zip(ld, lo):
            close(x.grad, y.grad, tol=1e-6)
    with pytest.raises(RuntimeError):
        npvp_amd.ops.l1_mean(ad, b.to(DEV)[..., :-1])


# ------------------------------------------------------------------------------- dropout (statistics + fwd/bwd replay)
def test_dropout_statistics_and_replay(K):
    from npvp_amd.ops import Drop
    dev = torch.device(DEV)
    R, N = 4096, 512
    x = torch.ones(R, N, device=DEV)
    d = Drop(0.1)
    y = K.drop_apply(x, d)
    keep = float((y != 0).float().mean())
    assert abs(keep - 0.9) < 0.003, keep
    assert abs(float(y.mean()) - 1.0) < 0.005
    assert torch.equal(y, K.drop_apply(x, d)), "same (seed, salt) must replay the same mask"
    assert not torch.equal(y, K.drop_apply(x, Drop(0.1))), "a new call site must draw a new mask"
    # GEMM epilogue dropout is the same stream as drop_apply with the same site
    w = torch.eye(N, device=DEV)
    d2 = Drop(0.1)
    assert torch.equal(K.linear_fwd(x, w, None, drop=d2) != 0, K.drop_apply(x, d2) != 0)
    # per-sample drop-path: whole groups of g1 rows share one decision
    dp = Drop(0.5, 1, 64, R // 64)
    z = K.drop_apply(x, dp).view(R // 64, -1)
    per = (z != 0).float().mean(1)
    assert bool(((per == 0) | (per == 1)).all()) and 0.3 < float(per.mean()) < 0.7
    # autograd replay: d/dx of sum(linear(x)) under dropout equals the forward mask pattern
    K.rng.begin_step(dev)
    xr = (torch.rand(256, N, device=DEV) + 0.5).requires_grad_(True)     # (no exact zeros: randn draws one per ~2e7 values)
    yy = K.linear(xr, w, None, None, Drop(0.3))
    (gx,) = torch.autograd.grad(yy.sum(), xr)
    assert torch.equal(gx != 0, yy != 0)


@pytest.mark.parametrize("R,g1,g2", [(4096, 512, 8), (8192, 64, 16), (2048, 64, 32)])
def test_drop_path_mask_rides_in_the_gemms(K, R, g1, g2):
    """The backward of a DropPath site: the row-group mask folded into the dgrad and weight-gradient GEMMs (a_drop: the scale of
    the staged rows / of the K-step, bias gradient included) against the masked copy of dy (npvp_drop_apply) + plain GEMMs."""
    from npvp_amd.ops import Drop
    if K.GEMM_PRECISION != 6:
        pytest.skip("a_drop is a feature of the fp16 kernels")
    dev = torch.device(DEV)
    N_out, K_in = 512, 512
    K.rng.manual_seed(5, dev)
    K.rng.begin_step(dev)
    dy = O.seeded_randn((R, N_out), 301).to(DEV)
    x = O.seeded_randn((R, K_in), 302).to(DEV)
    w = torch.nn.Parameter((O.seeded_randn((N_out, K_in), 303) / K_in ** 0.5).to(DEV))
    d = Drop(0.3, 1, g1, g2)
    dz, ad = K.masked_grad(dy, d, w)
    assert ad.on and dz is dy, "this shape must take the fused route"
    dx = K.linear_dgrad(dz, w, a_drop=ad)
    dw, db = K.linear_wgrad(dz, x, True, a_drop=ad)
    dzm = K.drop_apply(dy, d)
    groups = dzm.view(R // g1, -1)
    dead = int((groups.abs().sum(1) == 0).sum())
    assert 0 < dead < R // g1, "the mask must drop some groups and keep some"
    dx_r = K.linear_dgrad(dzm, w)
    dw_r, db_r = K.linear_wgrad(dzm, x, True)
    for a, b, n in ((dx, dx_r, "dx"), (dw, dw_r, "dw"), (db, db_r, "db")):
        e = float((a.double() - b.double()).norm() / b.double().norm())
        assert e < 1e-6, f"{n}: {e:.3e}"
    assert torch.equal(dx == 0, dx_r == 0), "dropped rows must be exactly zero in both routes"


@pytest.mark.parametrize("with_add,with_gamma", [(True, False), (False, True)])
def test_layernorm_and_positional_fuse_in_one_kernel(K, with_add, with_gamma):
    """npvp_ln_posfuse_fwd (frames of 64 token rows x 512 channels held in a block's registers) against npvp_layernorm_fwd followed
    by npvp_posfuse_fwd: LN output, fused output, both sets of statistics, both amax bounds."""
    N, T, P, C = 3, 4, 64, 512
    x = (0.7 + 1.3 * O.seeded_randn((N * T * P, C), 611)).to(DEV)
    lw, lb = (1 + 0.1 * O.seeded_randn((C,), 612)).to(DEV), (0.1 * O.seeded_randn((C,), 613)).to(DEV)
    add = O.seeded_randn((N, P, C), 614).to(DEV) if with_add else None
    beta = O.seeded_randn((T, P, C), 615).to(DEV)
    gamma = (0.3 * O.seeded_randn((T, P, C), 616)).to(DEV) if with_gamma else None
    keep = K.LN_POSFUSE_ONE_KERNEL
    try:
        K.LN_POSFUSE_ONE_KERNEL = True
        a = K._raw_ln_posfuse_fwd(x, lw, lb, 1e-5, add, beta, gamma, N, T)
        K.LN_POSFUSE_ONE_KERNEL = False
        b = K._raw_ln_posfuse_fwd(x, lw, lb, 1e-5, add, beta, gamma, N, T)
    finally:
        K.LN_POSFUSE_ONE_KERNEL = keep
    for u, v, n in zip(a, b, ["x1", "ln_stats", "fused", "pf_stats"]):
        e = float((u.double() - v.double()).norm() / v.double().norm())
        assert e < 2e-6, f"{n}: {e:.3e}"
    if K.GEMM_PRECISION == 6:
        for u, v in ((a[0], b[0]), (a[2], b[2])):
            assert abs(K.amax_of(u).read() - float(u.abs().max())) == 0.0 and abs(K.amax_of(v).read() - float(v.abs().max())) == 0.0


def test_mlpdwbn_backward_with_norm2_inside_the_fused_middle(K):
    """MlpDWBN backward, dropout ON: norm2's input gradient evaluated inside the fused middle's backward (frame sums + parameter
    gradients from npvp_frameln_act_bwd_pgrad, the dropout mask replayed element by element in npvp_mlpdw_mid_bwd_n2; dh2 never
    written) against the separate norm2 backward + npvp_mlpdw_mid_bwd, same inputs and the same mask stream."""
    if not K.mlpdwbn_fused_supported(6 * 64, 512, 2048, 512, 8, 8):
        pytest.skip("the fused MlpDWBN path needs a split GEMM mode")
    frames, T, C, hid, P = 6, 3, 512, 2048, 64
    R = frames * P
    mk = lambda shape, seed, sc=1.0, off=0.0: g((off + sc * O.seeded_randn(shape, seed)).requires_grad_())
    x, res = mk((R, C), 901), mk((R, C), 902)
    w1, b1 = mk((hid, C), 903, C ** -0.5), mk((hid,), 904, 0.1)
    n1w, n1b = mk((P * hid,), 905, 0.1, 1.0), mk((P * hid,), 906, 0.1)
    dww, dwb = mk((hid, 1, 3, 3), 907, 0.3), mk((hid,), 908, 0.1)
    n2w, n2b = mk((P * hid,), 909, 0.1, 1.0), mk((P * hid,), 910, 0.1)
    w2, b2 = mk((C, hid), 911, hid ** -0.5), mk((C,), 912, 0.1)
    n3w, n3b = mk((P * C,), 913, 0.1, 1.0), mk((P * C,), 914, 0.1)
    cot = O.seeded_randn((R, C), 915).to(DEV)
    leaves = [x, res, w1, b1, n1w, n1b, dww, dwb, n2w, n2b, w2, b2, n3w, n3b]
    dev = torch.device(DEV)

    def run(inside):
        keep = K.MID_BWD_N2
        K.MID_BWD_N2 = inside
        try:
            K.rng.manual_seed(77, dev)
            K.rng.begin_step(dev)
            y = K.mlpdwbn(*leaves, frames, T, 0.3, 0.2)
            return [y.detach()] + [t.detach() for t in torch.autograd.grad(y, leaves, cot)]
        finally:
            K.MID_BWD_N2 = keep

    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0])
    names = ["y", "dx", "dres", "dw1", "db1", "dn1w", "dn1b", "ddww", "ddwb", "dn2w", "dn2b", "dw2", "db2", "dn3w", "dn3b"]
    for u, v, n in zip(a, b, names):
        e = float((u.double() - v.double()).norm() / v.double().norm().clamp_min(1e-300))
        assert e < 2e-6, f"{n}: {e:.3e}"
    assert float((a[1] == 0).float().mean()) < 0.5 and float(a[9].abs().sum()) > 0


def test_attention_dropout_bwd_matches_finite_structure(K):
    """With attention dropout on, backward must replay forward's mask: check dV against a forward-mode identity
    (o is linear in v, so <cot, o(v)> = <dv, v>)."""
    from npvp_amd.ops import AttnCfg
    N, T, P, C = 1, 6, 64, 512
    q = torch.randn(N * T * P, C, device=DEV); k = torch.randn(N * T * P, C, device=DEV)
    v = torch.randn(N * T * P, C, device=DEV, requires_grad=True)
    cot = torch.randn(N * T * P, C, device=DEV)
    cfg = AttnCfg(1, N, P, 8, 0, T, T, 8, 0, 0.25)
    o = K.attn(q, k, v, cfg)
    (dv,) = torch.autograd.grad((o * cot).sum(), v)
    lhs, rhs = float((o * cot).sum()), float((dv * v).sum())
    assert abs(lhs - rhs) < 1e-3 * abs(lhs) + 1e-3, (lhs, rhs)


@pytest.mark.parametrize("frames", [160, 24, 1792])
def test_frame_ln_parameter_gradient_reduction_wide_sets(K, frames):
    """The second stage of the frame-LN parameter gradients: few partial rows (12 - 32 chunks) over 2 x 131 072 columns - the sets
    the float4 path of sum_rows takes (csrc/norm.hip sum_rows_wide).  The single launch against an fp64 sum, accumulate on and
    off, and the queued form (npvp_frameln_act_bwd_reduce_job + npvp_sum_rows_multi) bit-identical to it."""
    import ctypes
    from npvp_amd import ops
    from npvp_amd.sched import _ptr, _stream
    L = ops.lib()
    PF = 64 * 2048
    nbytes = L.npvp_frameln_act_bwd_workspace_bytes(frames, PF)
    head = frames * 8                                             # psum [frames][4 parts][2]
    nchunks = (nbytes // 4 - head) // (2 * PF)
    assert 2 <= nchunks <= 32 and head + nchunks * 2 * PF == nbytes // 4
    ws = torch.zeros(nbytes // 4, device=DEV)
    part = O.seeded_randn((nchunks, 2 * PF), 311 + frames).to(DEV)
    ws[head:] = part.reshape(-1)
    want = part.double().sum(0)
    base_w, base_b = O.seeded_randn((PF,), 5).to(DEV), O.seeded_randn((PF,), 6).to(DEV)
    for accumulate in (0, 1):
        dw, db = base_w.clone(), base_b.clone()
        assert L.npvp_frameln_act_bwd_reduce(_ptr(ws), _ptr(dw), _ptr(db), frames, PF, accumulate, _stream()) == 0
        ew = want[:PF] + (base_w.double() if accumulate else 0)
        eb = want[PF:] + (base_b.double() if accumulate else 0)
        assert float((dw.double() - ew).abs().max()) < 2e-5 and float((db.double() - eb).abs().max()) < 2e-5
        dw2, db2 = base_w.clone(), base_b.clone()
        job = ctypes.create_string_buffer(48)
        assert L.npvp_frameln_act_bwd_reduce_job(_ptr(ws), _ptr(dw2), _ptr(db2), frames, PF, accumulate, ctypes.addressof(job)) == 0
        assert L.npvp_sum_rows_multi(ctypes.addressof(job), 1, _stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(dw, dw2) and torch.equal(db, db2), "queued form differs from the single launch"


def test_library_exchange_entry_points(K):
    """include/npvp_hip.h npvp_dp_*: a communicator of ONE rank on this card (all a one-GPU box allows): id, init, an in-place
    all-reduce(mean) on a side stream, the compute stream ordered behind it by npvp_dp_wait, finalize - and the error returns
    around them.  The mean over one rank is the bucket itself."""
    import ctypes
    from npvp_amd import ops
    L = ops.lib()
    assert L.npvp_dp_world() == 0 and L.npvp_dp_rank() == -1
    x = O.seeded_randn((1 << 20,), 77).to(DEV)
    assert L.npvp_dp_allreduce_async(x.data_ptr(), x.numel(), None) == -1 and b"npvp_dp_init" in L.npvp_last_error()
    idb = ctypes.create_string_buffer(128)
    assert L.npvp_dp_unique_id(ctypes.addressof(idb)) == 0, L.npvp_last_error()
    assert any(idb.raw), "an all-zero communicator id"
    assert L.npvp_dp_init(1, 1, ctypes.addressof(idb)) == -1                   # rank out of range
    assert L.npvp_dp_init(0, 1, ctypes.addressof(idb)) == 0, L.npvp_last_error()
    try:
        assert L.npvp_dp_world() == 1 and L.npvp_dp_rank() == 0
        assert L.npvp_dp_init(0, 1, ctypes.addressof(idb)) == -1 and b"already" in L.npvp_last_error()
        side = torch.cuda.Stream()
        want = x.clone()
        y = x * 2.0                                                                 # produced on the compute stream ...
        side.wait_stream(torch.cuda.current_stream())                               # ... which the side stream is ordered after
        assert L.npvp_dp_allreduce_async(y.data_ptr(), y.numel(), side.cuda_stream) == 0, L.npvp_last_error()
        assert L.npvp_dp_wait(torch.cuda.current_stream().cuda_stream) == 0
        z = y * 0.5                                                                 # reads the reduced bucket on the compute stream
        torch.cuda.synchronize()
        assert torch.equal(z, want)
    finally:
        assert L.npvp_dp_finalize() == 0
    assert L.npvp_dp_world() == 0
    assert L.npvp_dp_wait(None) == -1


# ------------------------------------------------------------------------------- optimiser
def test_flat_adamw_matches_torch(K):
    import npvp_amd
    torch.manual_seed(0)
    class M(torch.nn.Module):
        def __init__(s):
            super().__init__()
            s.a = torch.nn.Linear(37, 19); s.transformer = torch.nn.Linear(19, 23)
    m1, m2 = M(), M()
    m2.load_state_dict(m1.state_dict())
    m2 = m2.to(DEV)
    o1 = torch.optim.AdamW(m1.parameters(), lr=1e-2)
    o2 = npvp_amd.FlatAdamW(m2, lr=1e-2, clip_module=m2.transformer, max_grad_norm=0.5)
    for it in range(3):
        x = O.seeded_randn((8, 37), 100 + it)
        o1.zero_grad(); o2.zero_grad()
        (m1.transformer(m1.a(x)) ** 2).sum().backward()
        (m2.transformer(m2.a(x.to(DEV))) ** 2).sum().backward()
        n1 = torch.nn.utils.clip_grad_norm_(m1.transformer.parameters(), 0.5)
        o1.step(); o2.step()
        close(o2.grad_norm().reshape(()), n1.reshape(()), tol=1e-5, what="grad norm")
        for (k1, p1), (k2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            close(p2, p1, tol=1e-5, what=f"step {it} {k1}")


@pytest.mark.parametrize("C,relu", [(512, True), (512, False), (256, True)])
def test_layernorm_writes_nchw(K, C, relu):
    """K9: the decoder's final LayerNorm (+ReLU) writing (N,T,C,H,W) directly, forward and backward, against
    LayerNorm + transpose through the separate kernels and against torch on the CPU."""
    from npvp_amd import ops
    N, T, H, W = 2, 3, 8, 8
    x = O.seeded_randn((N, T, H, W, C), 151)
    w = 1.0 + 0.2 * O.seeded_randn((C,), 152); b = 0.1 * O.seeded_randn((C,), 153)
    cot = O.seeded_randn((N, T, C, H, W), 154)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y = torch.nn.functional.layer_norm(xr, (C,), wr, br, 1e-5)
    y = (torch.relu(y) if relu else y).permute(0, 1, 4, 2, 3)
    ref = torch.autograd.grad((y * cot).sum(), [xr, wr, br])
    ins = [g(t.clone().requires_grad_()) for t in (x, w, b)]
    assert ops.layernorm_nchw_supported(ins[0], H, W)
    yg = ops.layernorm_nchw(ins[0], ins[1], ins[2], 1e-5, relu, N, T, H, W)
    got = torch.autograd.grad((yg * cot.to(DEV)).sum(), ins)
    close(yg, y.detach(), tol=1e-6, what="y")
    for a, b_, n in zip(got, ref, ["dx", "dw", "db"]):
        close(a, b_, tol=2e-6, what=n)
    ins2 = [g(t.clone().requires_grad_()) for t in (x, w, b)]
    y2 = ops.canonical_to_nchw(ops.layernorm(ins2[0], ins2[1], ins2[2], 1e-5, relu=relu), N, T, H, W)
    got2 = torch.autograd.grad((y2 * cot.to(DEV)).sum(), ins2)
    close(yg, y2, tol=1e-6, what="y vs two-kernel route")
    for a, b_, n in zip(got, got2, ["dx", "dw", "db"]):
        close(a, b_, tol=2e-6, what=n + " vs two-kernel route")


# ------------------------------------------------------------------------------- f16x3: amax slots, operand ranges, planes
def _f16(K):
    if K.GEMM_PRECISION != 6:
        pytest.skip("amax slots belong to the f16x3 mode")


def test_amax_slots_are_exact(K):
    """Every producer's amax slot holds exactly max|x| of the tensor it wrote (an integer atomic max of float bit patterns),
    views inherit their base's slot, and an in-place update voids a slot."""
    _f16(K)
    ops = K
    torch.manual_seed(3)
    dev = torch.device(DEV)

    def slot_of(t):
        tag = getattr(t, "_npvp_amax", None) or getattr(t._base, "_npvp_amax", None)
        assert tag is not None, "producer did not tag its output"
        return tag[0].read()

    x = torch.randn(1000, 512, device=dev) * 3.0
    w = torch.rand(512, device=dev) + 0.5; b = torch.randn(512, device=dev)
    y = ops.layernorm(x, w, b)
    assert slot_of(y) == float(y.abs().max())
    y2, st = ops._raw_ln_fwd(x, w, b, 1e-5)
    assert slot_of(y2) == float(y2.abs().max())
    # stand-alone reduction, strided rows, view lookup
    big = torch.randn(640, 1024, device=dev) * 1e-6
    s = ops.amax_of(big[:, :512])
    assert s.read() == float(big[:, :512].abs().max())
    s2 = ops.amax_of(big)
    assert ops.amax_of(big.view(-1, 1024)[5:]) is s2, "a view is bounded by its base's slot"
    big.mul_(2.0)
    assert ops.amax_of(big) is not s2, "an in-place update must void the slot"
    # drop_apply, posfuse, frame-LN, GEMM epilogue, attention
    d = ops.drop_apply(x, ops.Drop(0.3))
    assert slot_of(d) == float(d.abs().max())
    xf = torch.randn(6, 64 * 512, device=dev)
    beta = torch.randn(3, 64 * 512, device=dev)
    pf = ops.posfuse(xf, None, beta, None, 2, 3)
    assert slot_of(pf) == float(pf.abs().max())
    h = torch.randn(320, 2048, device=dev)
    a = ops.frameln_act(h, torch.rand(64 * 2048, device=dev) + 0.5, torch.randn(64 * 2048, device=dev), None, 5, p_drop=0.1)
    assert slot_of(a) == float(a.abs().max())
    wl = torch.nn.Parameter(torch.randn(1024, 512, device=dev) / 22.0)
    sl = ops.AmaxSlot.new(dev)
    yl = ops.linear_fwd(x[:768], wl, None, act=1, y_amax=sl)
    assert sl.read() == float(yl.abs().max())
    qk = torch.randn(4 * 64, 1024, device=dev); v = torch.randn(4 * 64, 512, device=dev)
    o = torch.empty_like(v)
    cfg = ops.AttnCfg(0, 4, 64, 8, 4, 0, 0, 8, 0, 0.0)
    ops._attn_fwd(qk[:, :512], qk[:, 512:], v, o, cfg)
    assert slot_of(o) == float(o.abs().max())
    go = torch.randn_like(o); dqk = torch.empty_like(qk); dv = torch.empty_like(v)
    ops._attn_bwd(qk[:, :512], qk[:, 512:], v, go, dqk[:, :512], dqk[:, 512:], dv, cfg, packed=dqk)
    assert slot_of(dqk) == float(dqk.abs().max()) and slot_of(dv) == float(dv.abs().max())


@pytest.mark.parametrize("scale", [1.0, 1e-8, 3e4, 1e-30])
def test_f16x3_operand_ranges(K, scale):
    """fp32-grade results wherever the operands live in fp32's range: gradient-sized (1e-8), large (3e4: above fp16's
    maximum once multiplied) and tiny operands, heavy-tailed rows; bar 1e-6 rel-L2 against fp64 (torch's fp32 GEMM: ~3e-7 ..
    6e-7 on these shapes)."""
    _f16(K)
    R = 4096
    g = torch.Generator(device=DEV).manual_seed(5)
    for N, K_ in ((512, 512), (2048, 512), (512, 2048)):
        w = torch.nn.Parameter(torch.randn(N, K_, device=DEV, generator=g) / math.sqrt(K_))
        x = torch.randn(R, K_, device=DEV, generator=g) * scale
        dy = torch.randn(R, N, device=DEV, generator=g) * torch.exp(1.5 * torch.randn(R, N, device=DEV, generator=g)) * scale
        close(K.linear_fwd(x, w, None), (x.double() @ w.double().T).float(), tol=1e-6, what=f"fwd {N}x{K_} scale {scale}")
        close(K.linear_dgrad(dy, w), (dy.double() @ w.double()).float(), tol=1e-6, what=f"dgrad {N}x{K_} scale {scale}")
        close(K.linear_wgrad(dy, x), (dy.double().T @ x.double()).float(), tol=1e-6, what=f"wgrad {N}x{K_} scale {scale}")


@pytest.mark.parametrize("M,N,K_", [(1024, 512, 512), (32768, 1024, 512)])      # 128 x 128 tiles / 128 x 256 tiles (>= 512 of them)
def test_heavy_tailed_rows_keep_their_precision(K, M, N, K_):
    """The per-ROW bar of north_star (1e-3 rel fp32), on operands whose token rows span twelve orders of magnitude and whose
    elements are log-normal inside a row (profiles/r03_gemm_bench_f16x3_heavy_tail_grads.txt: one scale per TENSOR left the
    smallest rows of a dgrad with 1e-3 errors at a rel-L2 of 3e-7).  The fp16 kernels' row guard recomputes a tile whose rows lie
    2^18 or more below the tensor's bound with per-row scales: every non-zero row of the product - forward, dgrad with a row-group
    mask, and every output-feature row of the weight gradient - must agree with an fp64 product to 1e-5, zero rows stay zero."""
    from golden_cases import max_row_rel_err
    if K.GEMM_PRECISION == 5:
        pytest.skip("bf16x3 is a 2^-16 arithmetic (opt-in weight-gradient mode): not held to the fp32-grade row bar")
    gen = torch.Generator().manual_seed(77)
    row_mag = 10.0 ** torch.empty(M, 1).uniform_(-12, 0, generator=gen)
    row_mag[5::97] = 0.0                                             # dropped samples: exact zero rows
    row_mag[0] = 1.0
    x = torch.randn(M, K_, generator=gen) * torch.exp(1.5 * torch.randn(M, K_, generator=gen)) * row_mag
    w = torch.randn(N, K_, generator=gen) / math.sqrt(K_)
    xg, wg = x.to(DEV), torch.nn.Parameter(w.to(DEV))
    ref = x.double() @ w.double().T
    y = K.linear_fwd(xg, wg, None)
    e, er = rel(y, ref), max_row_rel_err(y, ref, floor=0.0)
    assert e < 1e-5 and er < 1e-5, f"forward: rel-L2 {e:.2e}, worst row {er:.2e}"
    assert float(y[5::97].abs().max()) == 0.0
    # dgrad: the heavy-tailed operand is dy [M, N]; reduction over N
    dy = torch.randn(M, N, generator=gen) * torch.exp(1.5 * torch.randn(M, N, generator=gen)) * row_mag
    refd = dy.double() @ w.double()
    dx = K.linear_dgrad(dy.to(DEV), wg)
    e, er = rel(dx, refd), max_row_rel_err(dx, refd, floor=0.0)
    assert e < 1e-5 and er < 1e-5, f"dgrad: rel-L2 {e:.2e}, worst row {er:.2e}"
    # weight gradient: output features (columns of dy) spanning twelve orders of magnitude
    col_mag = 10.0 ** torch.empty(1, N).uniform_(-12, 0, generator=gen)
    col_mag[0, 3::41] = 0.0
    col_mag[0, 0] = 1.0
    dyc = torch.randn(M, N, generator=gen) * torch.exp(1.5 * torch.randn(M, N, generator=gen)) * col_mag
    xs = torch.randn(M, K_, generator=gen)
    refw = dyc.double().T @ xs.double()
    # the fp16 weight-gradient kernel DETECTS such features (a device counter) and the host re-runs the launch as bf16x6: at once in
    # strict mode (here), from the next step on in a training run (the counter is then read with the step's loss scalars)
    fp16_wgrad = K.GEMM_PRECISION == 6 and K._gemm_kernel_id(0, 0, N, K_, M, 6, False) == 6
    K.RangeGuard.reset()
    old_strict, K.RangeGuard.strict = K.RangeGuard.strict, True
    try:
        dw, db = K.linear_wgrad(dyc.to(DEV), xs.to(DEV), True)
        assert K.RangeGuard.events > 0 or not fp16_wgrad, "the fp16 weight-gradient kernel did not flag features 2^18 below the bound"
        into, into_b = torch.ones(N, K_, device=DEV), torch.ones(N, device=DEV)        # accumulation target untouched by the flagged try
        K.linear_wgrad(dyc.to(DEV), xs.to(DEV), True, into=into, into_b=into_b)
        close(into - 1.0, refw.float(), 1e-5, what="guarded weight gradient accumulated into a live slice", row_tol=1.0)
    finally:
        K.RangeGuard.strict = old_strict
    e, er = rel(dw, refw), max_row_rel_err(dw, refw, floor=0.0)
    assert e < 1e-5 and er < 1e-5, f"wgrad: rel-L2 {e:.2e}, worst row {er:.2e}"
    assert float(dw[3::41].abs().max()) == 0.0
    close(db, dyc.double().sum(0).float(), 1e-5, what="bias gradient beside a guarded weight gradient")
    # default mode: the launch itself is not repaired, the event is counted and arms the sticky bf16x6 fallback
    if fp16_wgrad:
        K.RangeGuard.reset()
        K.linear_wgrad(dyc.to(DEV), xs.to(DEV))
        assert K.RangeGuard.poll(torch.device(DEV)) > 0 and K.RangeGuard.fallback
        dw2 = K.linear_wgrad(dyc.to(DEV), xs.to(DEV))
        assert max_row_rel_err(dw2, refw, floor=0.0) < 1e-5, "sticky fallback: weight gradients after an event run as bf16x6"
        K.RangeGuard.reset()
        # well-ranged operands raise nothing
        K.linear_wgrad(torch.randn(M, N, device=DEV), xs.to(DEV))
        assert K.RangeGuard.poll(torch.device(DEV)) == 0 and not K.RangeGuard.fallback


def test_f16x3_nonfinite_operands_stay_nonfinite(K):
    """an inf / NaN in an operand must come out as inf / NaN (as in fp32), never as a finite number"""
    _f16(K)
    w = torch.nn.Parameter(torch.randn(512, 512, device=DEV) / 22.0)
    for bad in (float("inf"), float("nan")):
        x = torch.randn(1024, 512, device=DEV)
        x[7, 3] = bad
        y = K.linear_fwd(x, w, None)
        assert not bool(torch.isfinite(y[7]).all())
        assert bool(torch.isfinite(y[8:]).all()) or bad != bad      # (a NaN amax turns the scale to 1: other rows stay finite)


def test_weight_planes_follow_repointed_storage(K):
    """FlatBuffers re-points p.data; planes cached for the old storage must not be used (or refreshed) for the new
    one.  Forward -> optimiser #1 steps -> optimiser #2 (new flat buffers on the same module) steps with lr > 0 -> forward
    must see the weights optimiser #2 wrote."""
    if K.GEMM_PRECISION not in (4, 6):
        pytest.skip("planes exist in the split-precision modes")
    import npvp_amd
    torch.manual_seed(0)
    lin = torch.nn.Linear(512, 512).to(DEV)
    x = torch.randn(1024, 512, device=DEV)

    def fwd():
        return K.linear(x, lin.weight, lin.bias)

    def ref():
        return (x.double() @ lin.weight.detach().double().T + lin.bias.detach().double()).float()

    close(fwd(), ref(), tol=1e-5, what="before any optimiser")
    for round_ in range(2):
        opt = npvp_amd.FlatAdamW(lin, lr=1e-2)
        for _ in range(2):
            opt.zero_grad()
            fwd().square().mean().backward()
            opt.step()
        torch.cuda.synchronize()
        close(fwd(), ref(), tol=1e-5, what=f"after optimiser #{round_ + 1}")
