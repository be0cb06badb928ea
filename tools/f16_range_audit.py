"""Range audit of the fp16 two-term GEMMs on the tensors of a REAL training step (not synthetic operands):
for every forward / dgrad / weight-gradient launch that the fp16 kernels take during two steps of a c2-shaped run (BAIR NPVP-D,
2 + 28 frames, dropout 0.1 / drop-path 0.1 active, `--clips` clips) AFTER `--steps-before` optimiser steps (round 4: 200 - the
tensors of a model that has moved away from its initialisation, DropPath-zeroed samples included), record
  * the operand's dynamic range: its amax and how far its rows sit below it (rows whose own amax is more than 2^18 below the
    tensor's lose low-term bits; rows of dropped samples are exactly zero and are not counted);
  * the result's error against an fp64 product of the same operands (first 2048 rows): rel-L2 and the worst row.
  * epilogues with a bias, an activation (or its derivative) or a residual are evaluated too (reference built in fp64 from the same
    operands); only launches whose epilogue draws a dropout mask or accumulates into a live gradient are left out;
  * what the range guard did: weight-gradient range events (ops.RangeGuard), zero rows (dropped samples) per operand.
Usage: python tools/f16_range_audit.py [--clips 8] [--steps-before 200] > profiles/r04_f16_range_audit.txt"""
import sys, os, math, argparse, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import npvp_amd
from npvp_amd import ops
from npvp_amd._lib import lib
from npvp_amd.trainer import load_config

ap = argparse.ArgumentParser(); ap.add_argument("--clips", type=int, default=8); ap.add_argument("--steps-before", type=int, default=200)
args = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, To, Tp = args.clips, 2, 28
cfg = load_config(os.path.join(ROOT, "configs", "config_BAIR_VFP_NPVP-D.yaml"), B, To, Tp)
P = cfg["Predictor"]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = npvp_amd.build_predictor_from_cfg(npvp_amd.Predictor, P, To, Tp).to(dev).train()
opt = npvp_amd.FlatAdamW(model, lr=P["predictor_lr"], clip_module=model.transformer, max_grad_norm=P["max_grad_norm"])
ops.rng.manual_seed(1, dev)
g = torch.Generator().manual_seed(3)
past = torch.relu(torch.randn(B, To, 512, 8, 8, generator=g) * 0.1 + 0.05).to(dev)
fut = torch.relu(torch.randn(B, Tp, 512, 8, 8, generator=g) * 0.1 + 0.05).to(dev)
step = lambda: npvp_amd.predictor_train_step(model, opt, past, fut, P["lam_PF_L1"], P["KL_beta"], P["max_grad_norm"], sync=True)
ops.RangeGuard.reset()
for _ in range(max(1, args.steps_before)):      # un-audited optimiser steps (the first also does the first-use registrations)
    step()
events_before = ops.RangeGuard.events

rec = collections.defaultdict(list)
orig = ops.gemm
LAY = {(1, 1): "forward", (1, 0): "dgrad", (0, 0): "wgrad"}


def audited(a_kc, b_kc, M, N, K, A, lda, Bm, ldb, out, *a, **kw):
    r = orig(a_kc, b_kc, M, N, K, A, lda, Bm, ldb, out, *a, **kw)
    has_planes = kw.get("b_pre") is not None
    prec = kw.get("precision") or ops.GEMM_PRECISION
    kid = lib().npvp_gemm_kernel_id(a_kc, b_kc, M, N, K, prec, int(has_planes))
    drop_on = kw.get("drop", ops.NO_DROP).on or kw.get("a_drop", ops.NO_DROP).on
    plain = not kw.get("accumulate") and not drop_on          # (rowstats launches have a bias-only epilogue: evaluated)
    if kid not in (5, 6, 7):
        return r
    torch.cuda.synchronize()
    def rng_of(t):          # rows = the non-reduced index of the operand -> (amax, smallest non-zero row / amax, share below 2^-18, share of zero rows)
        ra = t.abs().amax(1)
        am = float(ra.max())
        nz = ra[ra > 0]
        lo = float(nz.min() / am) if nz.numel() else 1.0
        below = float((nz < am * 2.0 ** -18).float().mean()) if nz.numel() else 0.0
        return am, lo, below, 1.0 - nz.numel() / max(1, ra.numel())

    def gelu64(x):
        return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))

    def dgelu64(x):
        return 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)
    if a_kc:            # A [M][K] rows = token rows
        am, lo, below, zero = rng_of(A[:M])
        e = wr = float("nan")
        if plain:
            n = min(M, 2048)
            Wd = Bm.double()
            ref = A[:n].double() @ (Wd.T if b_kc else Wd)
            ref = ref * kw.get("alpha", 1.0)
            if kw.get("bias") is not None:
                ref = ref + kw["bias"].double()
            act = kw.get("act", 0)
            if act == 1:
                ref = gelu64(ref)
            elif act == 2:
                ref = torch.relu(ref)
            elif act == 3:
                ref = ref * dgelu64(kw["aux_in"][:n].double())
            elif act == 4:
                ref = ref * (kw["aux_in"][:n].double() > 0)
            if kw.get("residual") is not None:
                ref = ref + kw["residual"][:n].double()
            got = out[:n].double()
            e = float((got - ref).norm() / ref.norm().clamp_min(1e-300))
            rn = ref.norm(dim=1)
            ok = rn > 0
            wr = float(((got - ref).norm(dim=1)[ok] / rn[ok]).max()) if ok.any() else 0.0
        rec[(LAY[(a_kc, b_kc)], f"{M}x{N}x{K}")].append((am, lo, below, e, wr, zero))
    else:               # wgrad: A = dy [K][M], B = x [K][N]; the reduction runs over the rows
        am, lo, below, zero = rng_of(A[:K].T)         # per output-feature column of dy
        am2, lo2, below2, _ = rng_of(Bm[:K].T)
        e = wr = wr32 = float("nan")
        if plain:                       # (a launch that carries a row-group mask is compared where the mask is applied: tests/test_hip_ops.py)
            ref = A[:K].double().T @ Bm[:K].double()
            got = out.double()
            e = float((got - ref).norm() / ref.norm().clamp_min(1e-300))
            rn = ref.norm(dim=1); ok = rn > 0
            wr = float(((got - ref).norm(dim=1)[ok] / rn[ok]).max())
            t32 = (A[:K].T @ Bm[:K]).double()          # what an fp32 GEMM (rocBLAS through torch) makes of the same operands
            wr32 = float(((t32 - ref).norm(dim=1)[ok] / rn[ok]).max())
        rec[("wgrad", f"{M}x{N}x{K}")].append((min(am, am2), min(lo, lo2), max(below, below2), e, wr, float(((A[:K].abs().amax(1)) == 0).float().mean()), wr32))
    return r


ops.gemm = audited
ops.GradSink.enabled = False            # weight gradients into fresh tensors (so that they can be compared), on this stream
ops.WgradStream.enabled = False
for _ in range(2):
    step()
ops.gemm = orig
print(f"# fp16 two-term GEMMs on the tensors of two training steps after {args.steps_before} optimiser steps, BAIR NPVP-D {B} clips x (2 + 28), "
      f"dropout / drop-path 0.1")
print(f"# weight-gradient range events (a feature of dy 2^18 below its tensor's bound, ops.RangeGuard): {events_before} in the {args.steps_before} "
      f"steps before, {ops.RangeGuard.events - events_before + ops.RangeGuard.poll(dev)} in the two audited steps")
print("# per (layout, MxNxK): launches | operand amax range over launches | smallest (non-zero row amax / tensor amax) | largest share of")
print("# non-zero rows more than 2^18 below the tensor amax | largest share of exactly-zero rows (dropped samples / masked rows) |")
print("# rel-L2 vs fp64 (max) | worst row rel error (max)   [every epilogue but dropout-drawing / row-masked / accumulating ones is evaluated;")
print("# weight gradients: rows = output features, beside the worst row of torch's own fp32 GEMM on the same operands - long sums with cancellation]")
for (lay, shape), v in sorted(rec.items()):
    ams = [x[0] for x in v]; es = [x[3] for x in v if x[3] == x[3]]; ws = [x[4] for x in v if x[4] == x[4]]
    print(f"{lay:8s} {shape:20s} {len(v):4d} | amax {min(ams):.2e} .. {max(ams):.2e} | min row/tensor {min(x[1] for x in v):.1e} | "
          f"below 2^-18: {max(x[2] for x in v):.2%} | zero rows {max(x[5] for x in v):.1%} | evaluated {len(es):3d} | "
          f"rel-L2 {max(es) if es else float('nan'):.1e} | worst row {max(ws) if ws else float('nan'):.1e}"
          + (f" (torch fp32 GEMM on the same operands: {max(x[6] for x in v if x[6] == x[6]):.1e})" if lay == "wgrad" and any(x[6] == x[6] for x in v) else ""))
