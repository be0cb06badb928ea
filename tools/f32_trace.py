"""Localise a run-to-run difference: the golden suite's predictor-only training case, repeated; every GEMM call's operands and
output are check-summed (stream synchronised) and compared with the first run's.  Usage: python tools/f32_trace.py [f32|f16x3] [runs]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import npvp_amd as impl
from npvp_amd import ops
import golden_cases as GC

mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ops.set_gemm_precision(mode)
SYNC = os.environ.get("TRACE_SYNC", "1") == "1"       # 0: no check sums, no synchronisation - only the grad norms
if SYNC:
    ops.WgradStream.enabled = False
trace = []
_gemm = ops.gemm


def cs(t, rows, cols, ld):
    if t is None:
        return 0.0
    v = torch.as_strided(t, (rows, cols), (ld, 1))
    return float(v.double().sum()) + 3.0 * float(v.double().abs().sum())


def gemm(a_kc, b_kc, M, N, K, A, lda, B, ldb, out, **kw):
    ar, ac = (M, K) if a_kc else (K, M)
    br, bc = (N, K) if b_kc else (K, N)
    ia, ib = cs(A, ar, ac, lda), cs(B, br, bc, ldb)
    extra = sum(cs(kw.get(k), M, N, (kw.get(k).stride(0) if kw.get(k) is not None else 0)) for k in ("aux_in", "residual"))
    pre = cs(out, M, N, out.stride(0)) if kw.get("accumulate") else 0.0
    r = _gemm(a_kc, b_kc, M, N, K, A, lda, B, ldb, out, **kw)
    torch.cuda.synchronize()
    co = cs(out, M, N, out.stride(0))
    ccs = float(kw["colsum_a"].double().sum()) if kw.get("colsum_a") is not None else 0.0
    trace.append(((a_kc, b_kc, M, N, K, kw.get("act", 0), bool(kw.get("accumulate"))), ia, ib, extra, pre, co, ccs))
    return r


if SYNC:
    ops.gemm = gemm
mk = lambda m: impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
first = None
for r in range(runs):
    trace.clear()
    res = GC.case_train_step(impl, "cuda:0", "S", make_opt=mk)
    print(f"{mode} run {r}: grad_norm_0 {float(res['grad_norm_0']):.9e} grad_norm_1 {float(res['grad_norm_1']):.9e}  ({len(trace)} GEMM calls)", flush=True)
    if first is None:
        first = list(trace)
        continue
    shown = 0
    for i, (a, b) in enumerate(zip(first, trace)):
        if a != b:
            what = [n for n, x, y in zip(("shape", "A", "B", "aux/res", "C before", "C", "colsum"), a, b) if x != y]
            print(f"   call {i} {b[0]}: differs in {what}", flush=True)
            shown += 1
            if shown >= 6:
                break
