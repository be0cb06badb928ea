"""Which GEMM launches of a training step carry a dropout / DropPath mask in their epilogue (or on the rows of A), and which sites
mask a gradient copy (npvp_drop_apply): counts per (kind, shape) over ONE step of a BASELINE workload.
Usage: python tools/mask_sites.py [c2|c4|... a bench.py workload key] [clips]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import npvp_amd
from npvp_amd import ops
from npvp_amd.trainer import load_config

key = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg_file, name, B, To, Tp = bench.WORKLOADS[key]
if len(sys.argv) > 2:
    B = int(sys.argv[2])
dev = "cuda:0"
cfg = load_config(os.path.join(bench.ROOT, "configs", cfg_file), B, To, Tp)
P = cfg["Predictor"]
torch.manual_seed(0)
model = npvp_amd.build_predictor_from_cfg(npvp_amd.Predictor, P, To, Tp).to(dev).train()
opt = npvp_amd.FlatAdamW(model, lr=P["predictor_lr"], clip_module=model.transformer, max_grad_norm=P["max_grad_norm"])
ops.rng.manual_seed(0, dev)
past = torch.relu(torch.randn(B, To, 512, 8, 8) * 0.1 + 0.05).to(dev)
fut = torch.relu(torch.randn(B, Tp, 512, 8, 8) * 0.1 + 0.05).to(dev)
step = lambda: npvp_amd.predictor_train_step(model, opt, past, fut, P["lam_PF_L1"], P["KL_beta"], P["max_grad_norm"], sync=False)
step(); torch.cuda.synchronize()

counts = collections.Counter()
_gemm, _da = ops.gemm, ops.drop_apply
DIR = {(1, 1): "fwd", (1, 0): "dgrad", (0, 0): "wgrad"}


def gemm(a_kc, b_kc, M, N, K, *a, **kw):
    d, ad = kw.get("drop", ops.NO_DROP), kw.get("a_drop", ops.NO_DROP)
    kind = ("element mask" if d.mode == 0 else "row-group mask") if d.on else ("row-group mask on A" if ad.on else None)
    if kind:
        counts[(f"gemm {DIR.get((a_kc, b_kc), '?')}", kind, f"M={M} N={N} K={K}", f"act={kw.get('act', 0)}",
                "residual" if kw.get("residual") is not None else "")] += 1
    return _gemm(a_kc, b_kc, M, N, K, *a, **kw)


def drop_apply(x, d):
    counts[("drop_apply", "element mask" if d.mode == 0 else "row-group mask", f"[{x.shape[0]} x {x.shape[1]}]", "", "")] += 1
    return _da(x, d)


ops.gemm, ops.drop_apply = gemm, drop_apply
step(); torch.cuda.synchronize()
print(f"# {key}: {name}, {B} clips - masked launches of one training step")
for k, n in sorted(counts.items()):
    print(f"{n:4d}  " + "  ".join(x for x in k if x))
