"""Reproducer of the ROCm 7.2 graph packet-capture hazard (round 6) and check of the package's safe default.

A small predictor is trained for six steps twice: eagerly, and by replaying the step's HIP graph (trainer.GraphedTrainStep) with
something harmless done between the replays, chosen by BETWEEN:
    tiny        a 16-float tensor is allocated after the capture; ONE float of it is filled before every replay
    fill:<GiB>  a fresh buffer of that size is allocated, filled and freed before every replay
    pre         a buffer allocated BEFORE the capture is filled before every replay
    clone       the caller keeps `out["loss"].clone()` of every step (allocates 512 bytes per step)
    none        nothing
With the runtime's packet-capture path (NPVP_GRAPH_PACKET_CAPTURE=1; the ROCm default) `tiny`, `fill` and `clone` make ONE replay
compute a wrong update - whatever the bytes written are - and the losses leave the eager trajectory; `pre` and `none` are exact.
With DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (what `import npvp_amd` selects unless told otherwise) every variant is exact.
Prints the two loss sequences and "[graph_alloc_hazard] OK" when they agree to 1e-6.

    python tools/graph_alloc_hazard.py                      # package default: must print OK for every BETWEEN
    NPVP_GRAPH_PACKET_CAPTURE=1 BETWEEN=tiny python tools/graph_alloc_hazard.py     # shows the wrong step
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import npvp_amd as impl          # (first: it picks the runtime's replay mode before HIP initialises)
import torch
import golden_cases as GC
from oracle import ops as O

DEV = "cuda:0"
between = os.environ.get("BETWEEN", "tiny")
past = O.synth_features((2, 3, 512, 8, 8), 182).to(DEV); fut = O.synth_features((2, 4, 512, 8, 8), 183).to(DEV)
runs = {}
for graphed in (False, True):
    m = GC._small_predictor(impl, False, 181, DEV, evt_layers=1, dec_layers=1, dropout=0.1, drop_path=0.1)
    m.train()
    opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
    impl.ops.rng.manual_seed(77, torch.device(DEV))
    losses, kept = [], []
    if graphed:
        pre = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
        step = impl.GraphedTrainStep(m, opt, past, fut, 0.01, 1e-6, 1.0, warmup=1, prime=False)
        O.key_hashed_fill(m, 181)                   # (rewind: the constructor's warm-up step was a real one)
        opt.m.zero_(); opt.v.zero_(); opt.hyper[1:2].zero_()
        impl.ops.rng.manual_seed(77, torch.device(DEV))
        for i in range(6):
            torch.cuda.synchronize()
            if between == "tiny":
                if i == 0:
                    keep = torch.empty(16, dtype=torch.float32, device=DEV)
                keep[0:1].fill_(1.0)
            elif between.startswith("fill"):
                b = torch.empty(int(float(between.split(":")[1]) * (1 << 30)), dtype=torch.uint8, device=DEV)
                b.fill_(7)
                del b
            elif between == "pre":
                pre.fill_(7)
            out = step()
            torch.cuda.synchronize()
            losses.append(float(out["loss"]))
            if between == "clone":
                kept.append(out["loss"].clone())
    else:
        for i in range(6):
            out = impl.predictor_train_step(m, opt, past, fut, 0.01, 1e-6, 1.0, sync=False)
            torch.cuda.synchronize()
            losses.append(float(out["loss"]))
    runs[graphed] = losses
    print(("replayed" if graphed else "eager   "), " ".join(f"{l:.6f}" for l in losses), flush=True)
ok = all(abs(a - b) <= 1e-6 * abs(a) + 1e-9 for a, b in zip(runs[False], runs[True]))
print(f"[graph_alloc_hazard] packet capture {'ON' if impl.graph_packet_capture() else 'off'}, BETWEEN={between}: " + ("OK" if ok else "the replayed run LEFT the eager trajectory"), flush=True)
sys.exit(0 if ok else 1)
