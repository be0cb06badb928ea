"""cProfile of the host side of training steps (where does the Python time per step go?).
Usage: python tools/host_profile.py [steps] [c1|c4]"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import npvp_amd
from npvp_amd import ops
from npvp_amd.trainer import load_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wl = sys.argv[2] if len(sys.argv) > 2 else "c1"
B, To, Tp, cfgf = {"c1": (32, 10, 10, "config_KTH_VFP_NPVP-S.yaml"), "c4": (8, 4, 16, "config_KITTI_VFP_NPVP-D.yaml")}[wl]
cfg = load_config(os.path.join(ROOT, "configs", cfgf), B, To, Tp)
P = cfg["Predictor"]
dev = torch.device("cuda", 0)
model = npvp_amd.build_predictor_from_cfg(npvp_amd.Predictor, P, To, Tp).to(dev).train()
opt = npvp_amd.FlatAdamW(model, lr=1e-4, clip_module=model.transformer, max_grad_norm=1.0)
ops.rng.manual_seed(1, dev)
past = torch.relu(torch.randn(B, To, 512, 8, 8) * 0.1 + 0.05).to(dev)
fut = torch.relu(torch.randn(B, Tp, 512, 8, 8) * 0.1 + 0.05).to(dev)
step = lambda: npvp_amd.predictor_train_step(model, opt, past, fut, 0.01, 1e-8, 1.0, sync=False)
for _ in range(2):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
print(s.getvalue()[:5000])
