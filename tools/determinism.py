"""Two identical runs of 6 c1-like training steps (dropout active, gradient stream on) must produce bit-identical
parameters: every in-place gradient write is serialised on one stream in program order, reductions use fixed orders.
Usage: python tools/determinism.py"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import npvp_amd
from npvp_amd import ops
from npvp_amd.trainer import load_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, To, Tp = 8, 10, 10
cfg = load_config(os.path.join(ROOT, "configs", "config_KTH_VFP_NPVP-S.yaml"), B, To, Tp)
P = cfg["Predictor"]
dev = torch.device("cuda", 0)
digests = []
for run in range(2):
    torch.manual_seed(0)
    model = npvp_amd.build_predictor_from_cfg(npvp_amd.Predictor, P, To, Tp).to(dev).train()
    opt = npvp_amd.FlatAdamW(model, lr=1e-4, clip_module=model.transformer, max_grad_norm=1.0)
    ops.rng.manual_seed(1, dev)
    g = torch.Generator().manual_seed(5)
    past = torch.relu(torch.randn(B, To, 512, 8, 8, generator=g) * 0.1 + 0.05).to(dev)
    fut = torch.relu(torch.randn(B, Tp, 512, 8, 8, generator=g) * 0.1 + 0.05).to(dev)
    for _ in range(6):
        out = npvp_amd.predictor_train_step(model, opt, past, fut, 0.01, 1e-8, 1.0, sync=False)
    torch.cuda.synchronize()
    digests.append((hashlib.sha256(opt.flat_p.cpu().numpy().tobytes()).hexdigest()[:16], float(out["loss"])))
    print("run", run, digests[-1], flush=True)
assert digests[0] == digests[1], "non-deterministic training step"
print("bit-identical")
