"""Can library events (include/npvp_hip.h npvp_event_*) recorded INSIDE a stream capture be read for timing after a replay?
(torch.cuda.Event(external=True) is refused on ROCm: "External events are disallowed in rocm".)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from npvp_amd import ops
from npvp_amd.sched import ProbeEvent

dev = torch.device("cuda", 0)
x = torch.randn(8192, 8192, device=dev)
y = torch.empty_like(x)
evs = [ProbeEvent() for _ in range(4)]
evs[0].record(); torch.mm(x, x, out=y); evs[1].record()
torch.cuda.synchronize()
print(f"eager: mm {evs[0].elapsed_time(evs[1]):.3f} ms", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    evs[0].record()
    torch.mm(x, x, out=y)
    evs[1].record()
    y.mul_(2.0)
    evs[2].record()
    torch.mm(x, x, out=y)
    evs[3].record()
for r in range(3):
    g.replay()
    torch.cuda.synchronize()
    print(f"replay {r}: mm {evs[0].elapsed_time(evs[1]):.3f} ms, mul {evs[1].elapsed_time(evs[2]):.3f} ms, mm {evs[2].elapsed_time(evs[3]):.3f} ms", flush=True)
print("[ext_event_probe] OK")
