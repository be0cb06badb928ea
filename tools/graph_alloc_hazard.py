"""Reproducer of the ROCm 7.2 graph hazard of round 6 - memset NODES in a graph that the runtime replays from prepared AQL packets -
and check that the package's captured step is free of it.

A small predictor is trained for six steps twice: eagerly, and by replaying the step's HIP graph (trainer.GraphedTrainStep) with
something harmless done between the replays, chosen by BETWEEN:
    none        nothing
    tiny        a 16-float tensor is allocated after the capture; ONE float of it is filled before every replay
    tiny_sync   the same, and the device is synchronised between the fill and the replay
    fill:<GiB>  a fresh buffer of that size is allocated, filled and freed before every replay
    pre         a buffer allocated BEFORE the capture is filled before every replay
    clone       the caller keeps `out["loss"].clone()` of every step (allocates 512 bytes per step)
    inputs      the real caller: a NEW batch tensor every step, copied into the step's static inputs (step(past, fut))
Losses AND parameters of the two runs must be equal.  STOCHASTIC=1 takes the NPVP-S predictor (both runs draw the same noise: the
generator is re-seeded in front of the six steps, and a replay advances it exactly as the eager step does).

STEP_MEMSET=loss puts a memset node back into the captured step, as it had them until round 6: the feature loss as
torch.abs(a - b).mean() - torch's multi-block reduction clears a 4-byte semaphore with a memset.  With the runtime's prepared-packet
replay (NPVP_GRAPH_PACKET_CAPTURE=1; the ROCm default) that node is not executed reliably: with BETWEEN=tiny the loss
scalar of every replay after the first is never written (the parameters stay exact).  The other memset the step had - the 2 KB-per-
weight amax table that npvp_split_weights_f16 cleared with hipMemsetAsync, now a kernel - raced with the kernels that raise the
amaxes: the operand scales of the f16x3 GEMMs were cleared late and the PARAMETERS left the eager trajectory (bench.py replay_check
of the round's earlier tree, profiles/r06_graph_alloc_hazard.txt).  With DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (what `import npvp_amd` selects
unless told otherwise) every variant is exact, and so is the memset-free step in both modes.  (GraphedTrainStep refuses a step with
memset nodes under the prepared-packet mode; NPVP_ALLOW_GRAPH_MEMSETS=1 - set by this tool with STEP_MEMSET - lets the reproducer through.)

    python tools/graph_alloc_hazard.py                                              # package default: OK
    NPVP_GRAPH_PACKET_CAPTURE=1 BETWEEN=tiny python tools/graph_alloc_hazard.py     # memset-free step on the fast replay path: OK
    NPVP_GRAPH_PACKET_CAPTURE=1 BETWEEN=tiny STEP_MEMSET=loss python tools/graph_alloc_hazard.py    # shows the unwritten loss
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
step_memset = os.environ.get("STEP_MEMSET", "")
stochastic = os.environ.get("STOCHASTIC", "0") == "1"        # NPVP-S: both event encoders, KL term, one memcpy node in the graph
if step_memset:
    os.environ["NPVP_ALLOW_GRAPH_MEMSETS"] = "1"
    if "loss" in step_memset:
        class TorchL1:
            def __init__(self, norm_dim=None, lam=1.0):
                self.lam = lam

            def __call__(self, a, b):
                return torch.abs(b - a).mean() * self.lam
        impl.trainer.L1Loss = TorchL1
past = O.synth_features((2, 3, 512, 8, 8), 182).to(DEV); fut = O.synth_features((2, 4, 512, 8, 8), 183).to(DEV)
runs = {}
for graphed in (False, True):
    m = GC._small_predictor(impl, stochastic, 181, DEV, evt_layers=1, dec_layers=1, dropout=0.1, drop_path=0.1)
    m.train()
    opt = impl.FlatAdamW(m, lr=1e-4, clip_module=m.transformer, max_grad_norm=1.0)
    impl.ops.rng.manual_seed(77, torch.device(DEV))
    torch.cuda.manual_seed(4321)            # (NPVP-S: the reparameterisation noise comes from torch's generator)
    losses, kept = [], []
    if graphed:
        pre = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
        step = impl.GraphedTrainStep(m, opt, past, fut, 0.01, 1e-6, 1.0, warmup=1)
        O.key_hashed_fill(m, 181)                   # (rewind: the constructor's warm-up step was a real one)
        opt.m.zero_(); opt.v.zero_(); opt.hyper[1:2].zero_()
        impl.ops.rng.manual_seed(77, torch.device(DEV))
        torch.cuda.manual_seed(4321)
        for i in range(6):
            torch.cuda.synchronize()
            if between in ("tiny", "tiny_sync"):
                if i == 0:
                    keep = torch.empty(16, dtype=torch.float32, device=DEV)
                keep[0:1].fill_(1.0)
                if between == "tiny_sync":
                    torch.cuda.synchronize()
            elif between.startswith("fill"):
                b = torch.empty(int(float(between.split(":")[1]) * (1 << 30)), dtype=torch.uint8, device=DEV)
                b.fill_(7)
                del b
            elif between == "pre":
                pre.fill_(7)
            if between == "inputs":
                out = step(past.clone(), fut.clone())
            else:
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
    torch.cuda.synchronize()
    runs[(graphed, 'p')] = opt.flat_p.clone()
    if graphed:
        print(f"nodes of the captured step: {step.census}", flush=True)
    print(("replayed" if graphed else "eager   "), " ".join(f"{l:.6f}" for l in losses), flush=True)
same_p = bool(torch.equal(runs[(True, 'p')], runs[(False, 'p')]))
rel_p = float((runs[(True, 'p')] - runs[(False, 'p')]).norm() / runs[(False, 'p')].norm())
print(f"parameters after the six steps, replayed against eager: rel l2 diff {rel_p:.3e}, bit-equal {same_p}", flush=True)
ok = same_p and all(abs(a - b) <= 1e-6 * abs(a) + 1e-9 for a, b in zip(runs[False], runs[True]))
print(f"[graph_alloc_hazard] packet capture {'ON' if impl.graph_packet_capture() else 'off'}, BETWEEN={between}{' STEP_MEMSET=' + step_memset if step_memset else ''}{' STOCHASTIC' if stochastic else ''}: " + ("OK" if ok else "the replayed run LEFT the eager trajectory"), flush=True)
sys.exit(0 if ok else 1)
