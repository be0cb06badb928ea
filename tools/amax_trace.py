"""Which tensors still get their amax slot from the STAND-ALONE reduction (npvp_amax: one extra read of the tensor and one launch) instead
of from the kernel that produced them?  Counts per (shape, innermost npvp_amd call sites) over one training step of a workload.
    python tools/amax_trace.py [workload key, default c4]"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, npvp_amd
from npvp_amd import ops
from npvp_amd.trainer import load_config

key = sys.argv[1] if len(sys.argv) > 1 else "c4"
dev = torch.device("cuda:0")
cfg_file, name, B, To, Tp = bench.WORKLOADS[key]
cfg = load_config(os.path.join(ROOT, "configs", cfg_file), B, To, Tp)
P = cfg["Predictor"]
model = npvp_amd.build_predictor_from_cfg(npvp_amd.Predictor, P, To, Tp).to(dev).train()
opt = npvp_amd.FlatAdamW(model, lr=P["predictor_lr"], clip_module=model.transformer, max_grad_norm=P["max_grad_norm"])
past = torch.relu(torch.randn(B, To, 512, 8, 8) * 0.1 + 0.05).to(dev)
fut = torch.relu(torch.randn(B, Tp, 512, 8, 8) * 0.1 + 0.05).to(dev)
step = lambda: npvp_amd.predictor_train_step(model, opt, past, fut, P["lam_PF_L1"], P["KL_beta"], P["max_grad_norm"], sync=False)
torch.autograd.set_multithreading_enabled(False)
for _ in range(2):
    step()
sites = collections.Counter()
orig = ops.amax_of


def traced(t, slot=None):
    if slot is None:
        hit = False
        for cand in (t, t._base):
            tag = getattr(cand, "_npvp_amax", None) if cand is not None else None
            if tag is not None and tag[1] == cand._version:
                hit = True
        if not hit:
            fr = [f for f in traceback.extract_stack(limit=12) if "npvp_amd" in f.filename and "amax_of" not in f.name]
            sites[(tuple(t.shape), " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in fr[-3:]))] += 1
    return orig(t, slot)


ops.amax_of = traced
step()
torch.cuda.synchronize()
for (shape, site), n in sites.most_common(30):
    print(f"{n:4d}  {shape}  {site}")
