"""The engine-clock state of a replayed step (round 6).  The same HIP graph of the c2 step replays at 237 ms or at 214 ms per step
depending on what ran before it: after ONE long hipMemsetAsync (torch `buf.zero_()` over a whole >= ~16 GiB buffer) rocm-smi reads
sclk ~2 270 MHz instead of ~1 970 MHz, the MFMA GEMMs run 19 % faster, and the state holds for as long as the queue never drains
(60 replays = 13 s checked).  This tool replays a workload's graph in event-timed loops after different primers and samples the
clocks beside it (tools/clock_watch.sh), so that the effect and its trigger are on record:

    NPVP_GRAPH_PACKET_CAPTURE=1 bash tools/clock_watch.sh gpurun_out/r06/clocks.txt -- python tools/clock_prime_probe.py [workload] [eager]

(the effect was measured on the runtime's packet-capture replay path, which `import npvp_amd` switches off by default because its
arithmetic is broken - profiles/r06_graph_alloc_hazard.txt; with the safe path the primer does nothing)
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import npvp_amd
from npvp_amd import ops
from npvp_amd.trainer import load_config
import bench

key = sys.argv[1] if len(sys.argv) > 1 else "c2"
eager = len(sys.argv) > 2 and sys.argv[2] == "eager"
cfg_file, name, B, To, Tp = bench.WORKLOADS[key]
dev = torch.device("cuda", 0)
cfg = load_config(os.path.join(ROOT, "configs", cfg_file), B, To, Tp)
P = cfg["Predictor"]
torch.manual_seed(3047)
model = npvp_amd.build_predictor_from_cfg(npvp_amd.Predictor, P, To, Tp).to(dev).train()
opt = npvp_amd.FlatAdamW(model, lr=P["predictor_lr"], clip_module=model.transformer, max_grad_norm=P["max_grad_norm"])
g = torch.Generator().manual_seed(3047)
past = torch.relu(torch.randn(B, To, 512, 8, 8, generator=g) * 0.1 + 0.05).to(dev)
fut = torch.relu(torch.randn(B, Tp, 512, 8, 8, generator=g) * 0.1 + 0.05).to(dev)
args = (P["lam_PF_L1"], P["KL_beta"], P["max_grad_norm"])
if eager:
    step = lambda: npvp_amd.predictor_train_step(model, opt, past, fut, *args, sync=False)
    for _ in range(3):
        step()
else:
    gs = npvp_amd.GraphedTrainStep(model, opt, past, fut, *args, prime=False)
    step = lambda: gs()
torch.cuda.synchronize()
print(f"[clock_prime_probe] {name}: {'eager two-stream step' if eager else 'single-stream graph replay'}", flush=True)


def loop(tag, n=8):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for i in range(n):
        evs[i].record()
        step()
    evs[n].record()
    torch.cuda.synchronize()
    print(f"[clock_prime_probe] t={time.time():.1f} {tag}: device ms per step: " + " ".join(f"{evs[i].elapsed_time(evs[i + 1]):.1f}" for i in range(n)), flush=True)


G = 1 << 30
long_n = int(os.environ.get("PRIME_PROBE_LONG", "60"))
loop("baseline")
for gib in (4, 8, 10, 12, 16, 24):
    loop("reset")
    # zero_() of a tensor that owns its whole storage is ONE hipMemsetAsync (on a slice view it would be a fill kernel); the buffer
    # goes straight back to the DEVICE afterwards, so that the probe never oversubscribes memory (round 6: with ~260 GiB in use the
    # driver evicted part of the graph's pool and the same graph ran 12 % slower at the same clock)
    buf = torch.empty(gib * G, dtype=torch.uint8, device=dev)
    buf.zero_()
    loop(f"after hipMemsetAsync of {gib} GiB")
    del buf
    torch.cuda.empty_cache()
    loop("the loop after")
big = torch.empty(24 * G, dtype=torch.uint8, device=dev)
loop("reset")
big.zero_()
loop(f"after hipMemsetAsync of 24 GiB, then {long_n} steps", n=long_n)
torch.cuda.synchronize()
time.sleep(1.0)
loop("after 1 s of idle")
big[:12 * G].zero_()
loop("after a FILL KERNEL over 12 GiB (zero_ of a slice view)")
from npvp_amd.sched import prime_clocks
print("[clock_prime_probe] sched.prime_clocks ->", prime_clocks(dev), "bytes", flush=True)
loop("after sched.prime_clocks")
print("[clock_prime_probe] done", flush=True)
