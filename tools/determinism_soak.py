"""Run-to-run bitwise reproducibility soak of the predictor train step (GPU).

    B=32 RUNS=12 python tools/determinism_soak.py

Builds the same model RUNS times from the same seeds, takes 3 training steps on the same synthetic features and
prints the number of distinct SHA-256 digests of the flat parameter buffer (1 = reproducible).  DESIGN.md section 7
quotes its results; tests/test_hip_golden.py::test_training_step_is_bitwise_deterministic is the small in-suite version.
"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                     # noqa: E402
import npvp_amd                  # noqa: E402
from npvp_amd import ops         # noqa: E402

DEV = "cuda:0"
B, To, Tp = int(os.environ.get("B", "32")), 10, 10
RUNS = int(os.environ.get("RUNS", "4"))
h = torch.linspace(0, 7, 8)
g = torch.Generator().manual_seed(1)
past = torch.relu(torch.randn(B, To, 512, 8, 8, generator=g) * 0.1 + 0.05).to(DEV)
fut = torch.relu(torch.randn(B, Tp, 512, 8, 8, generator=g) * 0.1 + 0.05).to(DEV)
digests = []
for r in range(RUNS):
    torch.manual_seed(0)
    m = npvp_amd.Predictor(8, 8, 20, h, h, torch.linspace(0, 9, 10), torch.linspace(10, 19, 10), 512, 'Add', 'layer', 256, 1,
                           True, 8, evt_former=True, learn_evt_token=False, evt_former_num_layers=4).to(DEV)
    m.train()
    opt = npvp_amd.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
    ops.rng.manual_seed(9, torch.device(DEV))
    for s in range(3):
        npvp_amd.predictor_train_step(m, opt, past, fut, 0.01, 1e-8, 1.0, sync=False)
    torch.cuda.synchronize()
    digests.append(hashlib.sha256(opt.flat_p.cpu().numpy().tobytes()).hexdigest()[:8])
    del m, opt
print(f"B={B} runs={RUNS}: {len(set(digests))} distinct digest(s): {digests}")
