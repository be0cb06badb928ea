"""How far does the gradient stream lag behind the backward pass?  Records an event on the main stream when the backward
chain has been enqueued completely and one on the gradient stream after its last kernel; the difference is the tail the
optimiser has to wait for.  Usage: python tools/grad_tail.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import npvp_amd
from npvp_amd import ops
from npvp_amd.trainer import load_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, To, Tp = 32, 10, 10
cfg = load_config(os.path.join(ROOT, "configs", "config_KTH_VFP_NPVP-S.yaml"), B, To, Tp)
P = cfg["Predictor"]
dev = torch.device("cuda", 0)
model = npvp_amd.build_predictor_from_cfg(npvp_amd.Predictor, P, To, Tp).to(dev).train()
opt = npvp_amd.FlatAdamW(model, lr=1e-4, clip_module=model.transformer, max_grad_norm=1.0)
ops.rng.manual_seed(1, dev)
past = torch.relu(torch.randn(B, To, 512, 8, 8) * 0.1 + 0.05).to(dev)
fut = torch.relu(torch.randn(B, Tp, 512, 8, 8) * 0.1 + 0.05).to(dev)
for _ in range(3):
    npvp_amd.predictor_train_step(model, opt, past, fut, 0.01, 1e-8, 1.0, sync=False)
torch.cuda.synchronize()
ev = {}
orig_join = ops.WgradStream.join.__func__


def join(cls):
    if cls._pending is not None:
        d, side = cls._pending
        ev["main_done"] = torch.cuda.Event(enable_timing=True); ev["main_done"].record(torch.cuda.current_stream(d))
        ev["side_done"] = torch.cuda.Event(enable_timing=True); ev["side_done"].record(side)
    orig_join(cls)


ops.WgradStream.join = classmethod(join)
lags = []
for _ in range(5):
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    npvp_amd.predictor_train_step(model, opt, past, fut, 0.01, 1e-8, 1.0, sync=False)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()
    torch.cuda.synchronize()
    lags.append((e0.elapsed_time(ev["main_done"]), e0.elapsed_time(ev["side_done"]), e0.elapsed_time(e1)))
for a, b, c in lags:
    print(f"backward chain done at {a:7.2f} ms, gradient stream done at {b:7.2f} ms (tail {b - a:6.2f} ms), step {c:7.2f} ms")
