"""Repeat the from-pixels training-step case of the golden suite in one GEMM mode; print grad_norm of every run with its error
against the golden vector, and - against the first run - the parameters whose gradient differs: run-to-run differences mean a race
or a read of uninitialised memory.  Usage: python tools/f32_repeat.py [f32|f16x3] [runs]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import npvp_amd as impl
import golden_cases as GC

mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
impl.ops.set_gemm_precision(mode)
gold = GC.load("train_step_full_S")
opts = []


def mk(m):
    o = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
    opts.append((o, m))
    return o


first = None
for r in range(runs):
    res = GC.case_full_step(impl, impl, "cuda:0", make_opt=mk, device_layout=None)
    gn = float(res["grad_norm"]); g = float(gold["grad_norm"])
    o, m = opts[-1]
    names = {id(p): n for n, p in m.named_parameters()}
    buf = o.buf
    grads = {names.get(id(p), "?"): buf.flat_g[off:off + n].clone() for p, (off, n) in zip(buf.params, buf.offsets)}
    print(f"{mode} run {r}: grad_norm {gn:.9e} (golden rel {abs(gn - g) / g:.3e})", flush=True)
    if first is None:
        first = grads
    else:
        for n, gt in grads.items():
            d = float((gt - first[n]).norm()); b = float(first[n].norm())
            if d > 1e-6 * max(b, 1e-20):
                print(f"      {n:60s} |g| {b:.4e}  |g - g_run0| {d:.3e}  ({d / max(b, 1e-30):.2e})", flush=True)
