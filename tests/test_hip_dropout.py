"""GPU: train-mode parity WITH dropout and drop-path active (the benchmark runs with both at 0.1).

The HIP kernels draw their masks from a counter hash, the reference from torch's RNG - they cannot match bit for bit.
So the masks the kernels actually drew are read back (ops.DropRecorder replays every site through npvp_drop_apply on
ones) and INJECTED into the oracle at the corresponding call sites (nn.Dropout / F.dropout, the attention-probability
dropout inside nn.MultiheadAttention's slow path, DropPath): both sides then compute the same function and must agree
to the usual 1e-4 (bar 1e-3), forward and backward.  Also asserted: the structure of the reference's DropPath keying -
per SAMPLE at the spatial / conv-FFN sites (ref/models/VidHRFormer.py:88,91,212,214,243) and per TIME-STEP at the
enc-dec site, where the reference's tensor is (T2, N*H*W, C) (ref :239,513-525)."""
import pytest
import torch
import torch.nn.functional as F

import golden_cases as GC
from oracle import ops as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _Inject:
    """pops the recorded masks in call order inside the oracle"""

    def __init__(self, masks):
        self.masks, self.i = masks, 0

    def next(self, kind, numel):
        assert self.i < len(self.masks), "the oracle asked for more dropout sites than the HIP path recorded"
        (drop, k, count), m = self.masks[self.i]
        self.i += 1
        assert k == kind and count == numel, f"site {self.i - 1}: HIP recorded ({k}, {count}), oracle wants ({kind}, {numel})"
        return m

    def dropout(self, x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        return x * self.next("elem", x.numel()).view(x.shape)

    def drop_path(self, x, p, training, dim):
        if p == 0.0 or not training:
            return x
        shape = [1] * x.ndim
        shape[dim] = x.shape[dim]
        return x * self.next("group", x.shape[dim]).view(shape)


def _run_pair(make_hip, make_ref, run, seed):
    """-> (hip outputs, oracle outputs, recorded sites)"""
    import npvp_amd
    from npvp_amd import ops
    import oracle.model as OM
    dev = torch.device(DEV)
    ops.set_gemm_precision("bf16x6")
    hip, ref = make_hip(), make_ref()
    O.key_hashed_fill(ref, seed); O.key_hashed_fill(hip, seed)
    hip = hip.to(DEV).train(); ref.train()
    ops.rng.manual_seed(4242, dev)
    ops.DropRecorder.sites = []
    try:
        out_h = run(hip, DEV)
        sites = ops.DropRecorder.sites
    finally:
        ops.DropRecorder.sites = None
    masks = [(s, ops.DropRecorder.mask(s, dev).cpu()) for s in sites]
    inj = _Inject(masks)
    old_f, old_dp = F.dropout, OM._drop_path
    torch.nn.functional.dropout, OM._drop_path = inj.dropout, inj.drop_path
    try:
        out_r = run(ref, "cpu")
    finally:
        torch.nn.functional.dropout, OM._drop_path = old_f, old_dp
    assert inj.i == len(masks), (f"the oracle consumed {inj.i} of {len(masks)} recorded dropout sites: "
                                 f"{[(s[1], s[2], s[0].salt) for s, _ in masks]}")
    return out_h, out_r, masks


def _check(out_h, out_r, names, tol=1e-4):
    for a, b, n in zip(out_h, out_r, names):
        e = GC.rel_err(a, b)
        assert e < tol, f"{n}: rel-L2 {e:.3e} with injected dropout masks"


def test_block_enc_with_dropout():
    import npvp_amd, oracle
    N, T = 3, 4
    mk = lambda pkg: (lambda: pkg.VidHRFormerBlockEnc(8, 8, 512, 8, window_size=4, dropout=0.1, drop_path=0.4))
    x0 = O.seeded_randn((N, T, 8, 8, 512), 301); cot = O.seeded_randn((N, T, 8, 8, 512), 302)
    beta = 0.5 * O.seeded_randn((T * 64, 512), 303)

    def run(m, d):
        fz = (npvp_amd if d != "cpu" else oracle).PosFeatFuser(512, "layer")
        x = x0.to(d).requires_grad_()
        y = m(x, (beta.to(d), None), fz)
        (gx,) = torch.autograd.grad((y * cot.to(d)).sum(), x)
        return y.detach().cpu(), gx.cpu()

    out_h, out_r, masks = _run_pair(mk(npvp_amd), mk(oracle), run, 31)
    _check(out_h, out_r, ["y", "gx"])
    kinds = [(s[1], s[2]) for s, _ in masks]
    # A1, DP1(sample), M1, M2, DP2(sample), A2, D1, F1, F2
    assert kinds == [("elem", N * T * 4 * 8 * 16 * 16), ("group", N), ("elem", N * T * 64 * 2048), ("elem", N * T * 64 * 512),
                     ("group", N), ("elem", N * 64 * 8 * T * T), ("elem", N * T * 64 * 512), ("elem", N * T * 64 * 1024),
                     ("elem", N * T * 64 * 512)], kinds
    for (s, m) in masks:                      # keep-scale values are 0 or 1/(1-p)
        assert all(v == 0.0 or abs(v - 1.0 / (1.0 - s[0].p)) < 1e-6 for v in m.unique().tolist())
    assert 0.85 < float((masks[2][1] != 0).float().mean()) < 0.95


def test_block_dec_with_dropout_and_time_step_drop_path():
    import npvp_amd, oracle
    N, T1, T2 = 2, 2, 5
    mk = lambda pkg: (lambda: pkg.VidHRFormerBlockDecNAR(8, 8, 512, 8, window_size=4, dropout=0.1, drop_path=0.4))
    tgt0 = O.seeded_randn((N, T2, 8, 8, 512), 311); qe0 = 0.5 * O.seeded_randn((N, 8, 8, 512), 312)
    mem0 = O.seeded_randn((N, T1, 8, 8, 512), 313); cot = O.seeded_randn((N, T2, 8, 8, 512), 314)
    tb = 0.5 * O.seeded_randn((T2 * 64, 512), 315); mb = 0.5 * O.seeded_randn((T1 * 64, 512), 316)

    def run(m, d):
        fz = (npvp_amd if d != "cpu" else oracle).PosFeatFuser(512, "layer")
        tgt, qe, mem = tgt0.to(d).requires_grad_(), qe0.to(d).requires_grad_(), mem0.to(d).requires_grad_()
        y = m(tgt, qe, mem, (mb.to(d), None), (tb.to(d), None), fz)
        g = torch.autograd.grad((y * cot.to(d)).sum(), [tgt, qe, mem])
        return [y.detach().cpu()] + [t.cpu() for t in g]

    out_h, out_r, masks = _run_pair(mk(npvp_amd), mk(oracle), run, 32)
    _check(out_h, out_r, ["y", "g_tgt", "g_query_evt", "g_memory"])
    kinds = [(s[1], s[2]) for s, _ in masks]
    # SLMHSA (A1, DP sample), SpatialFFN (M1, M2, DP sample), temporal (A2, D1), FFN (F1, F2), EncDec (A3, DP per TIME-STEP),
    # SpatialFFN1 (M3, M4, DP sample)
    want = [("elem", N * T2 * 4 * 8 * 16 * 16), ("group", N), ("elem", N * T2 * 64 * 2048), ("elem", N * T2 * 64 * 512), ("group", N),
            ("elem", N * 64 * 8 * T2 * T2), ("elem", N * T2 * 64 * 512), ("elem", N * T2 * 64 * 1024), ("elem", N * T2 * 64 * 512),
            ("elem", N * 64 * 8 * T2 * T1), ("group", T2), ("elem", N * T2 * 64 * 2048), ("elem", N * T2 * 64 * 512), ("group", N)]
    assert kinds == want, kinds
    # the enc-dec drop-path site drops whole time-steps for every sample and pixel: its Drop is keyed (row / P) % T2
    d = masks[10][0][0]
    assert (d.mode, d.g1, d.g2) == (1, 64, T2)
    d = masks[1][0][0]
    assert (d.mode, d.g1, d.g2) == (1, T2 * 64, N)          # per sample elsewhere


def test_enc_dec_drop_path_is_constant_over_samples_and_pixels():
    """Direct kernel-level check of the per-time-step keying: out-proj GEMM epilogue with Drop(p, 1, P, T2) on ones."""
    from npvp_amd import ops
    dev = torch.device(DEV)
    N, T2, P, C = 3, 6, 64, 512
    ops.rng.manual_seed(7, dev)
    d = ops.Drop(0.5, 1, P, T2)
    y = ops.drop_apply(torch.ones(N * T2 * P, C, device=DEV), d).view(N, T2, P, C)
    per_t = y[0, :, 0, 0]
    assert bool((y == per_t.view(1, T2, 1, 1)).all()), "mask must depend on the time-step only"
    assert 0 < int((per_t != 0).sum()) < T2 or T2 < 4
    w = torch.eye(C, device=DEV)
    z = ops.linear_fwd(torch.ones(N * T2 * P, C, device=DEV), w, None, drop=d).view(N, T2, P, C)
    assert torch.equal(z != 0, y != 0), "GEMM-epilogue drop-path must replay the same per-time-step mask"


@pytest.mark.parametrize("g1,g2", [(1, 77), (7, 13), (64, 5), (448, 3), (1792, 64), (100, 1), (3, 1000)])
def test_row_group_keys_follow_the_integer_division(g1, g2):
    """A DropPath site keys its mask by (row / g1) % g2.  The kernels take that quotient by a multiply-shift with host-side magic
    numbers (csrc/common.h div_magic: two 64-bit divisions per row were most of a GEMM epilogue's instructions); here every row's
    decision - through npvp_drop_apply and through the epilogue of every GEMM generation - must be the decision of its group in
    the plain integer arithmetic."""
    import numpy as np
    from npvp_amd import ops
    dev = torch.device(DEV)
    ops.rng.manual_seed(977, dev)
    R = 5000 if g1 * g2 < 5000 else 8192
    d = ops.Drop(0.3, 1, g1, g2)
    per_group = ops.DropRecorder.mask((d, "group", g2), dev).cpu().numpy()          # decision of group g = row g at g1 = 1
    assert 0 < (per_group == 0).sum() < g2 or g2 < 8
    want = per_group[(np.arange(R) // g1) % g2]
    got = ops.drop_apply(torch.ones(R, 4, device=DEV), d)[:, 0].cpu().numpy()
    assert np.array_equal(got, want), "npvp_drop_apply"
    x = torch.randn(R, 512, device=DEV)
    w = torch.randn(512, 512, device=DEV) * 0.05
    res = torch.randn(R, 512, device=DEV)
    for mode in ("f16x3", "bf16x6", "f32"):
        ops.set_gemm_precision(mode)
        try:
            plain = ops.linear_fwd(x, w, None)
            y = ops.linear_fwd(x, w, None, residual=res, drop=d)
        finally:
            ops.set_gemm_precision("f16x3")
        ref = plain * torch.from_numpy(want).to(DEV)[:, None] + res
        assert GC.rel_err(y, ref) < 1e-6, f"{mode}: masked epilogue differs from mask x plain output"


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_mask_statistics(p):
    """The counter hash behind every mask (csrc/common.h rng_u32: two keyed finalizer rounds per element): keep rate, independence
    between sites (salts) and steps (seeds), and no lagged structure along a row or down a column of the [R, 512] / [R, 2048]
    tensors it masks - each within 4.5 sigma of what independent Bernoulli draws give over 2^22 elements."""
    import numpy as np
    from npvp_amd import ops
    dev = torch.device(DEV)
    n = 1 << 22
    sig = 1.0 / np.sqrt(n)
    masks = []
    for seed in (4242, 4243, 99991):
        ops.rng.manual_seed(seed, dev)
        for _ in range(3):
            d = ops.Drop(p, 0)                                  # a new site: the next salt
            m = ops.DropRecorder.mask((d, "elem", n), dev)
            masks.append((m != 0).double().cpu().numpy())
    for m in masks:
        assert abs(m.mean() - (1 - p)) < 4.5 * np.sqrt(p * (1 - p)) * sig, f"keep rate {m.mean():.5f} at p = {p}"
    z = [(m - m.mean()) / m.std() for m in masks]
    for i in range(len(z)):
        for j in range(i + 1, len(z)):
            assert abs((z[i] * z[j]).mean()) < 4.5 * sig, f"masks {i} and {j} are correlated: {(z[i] * z[j]).mean():.2e}"
    for lag in (1, 2, 3, 4, 8, 64, 512, 2048, 4096, 2048 * 64):
        for k in (0, 4, 8):
            r = (z[k][:-lag] * z[k][lag:]).mean()
            assert abs(r) < 4.5 * sig, f"lag {lag}: autocorrelation {r:.2e}"
    # counts per row of a [8192, 512] view and per column: binomial variance
    a = masks[0].reshape(8192, 512)
    assert 0.9 < a.sum(1).var() / (512 * p * (1 - p)) < 1.1
    assert 0.75 < a.sum(0).var() / (8192 * p * (1 - p)) < 1.25
