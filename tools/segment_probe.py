"""Feasibility probe for trainer.SegmentedTrainStep: can a HIP stream capture be ENDED and a new one BEGUN from inside the autograd
engine's worker thread (capture_error_mode="relaxed"), all segments sharing one memory pool, with host actions between the segments
at replay time?  Prints the thread ids, the per-segment replay results and the difference to the eager gradients."""
import threading
import torch

dev = torch.device("cuda", 0)
torch.manual_seed(0)
w1 = torch.randn(512, 512, device=dev, requires_grad=True)
w2 = torch.randn(512, 512, device=dev, requires_grad=True)
xin = torch.randn(256, 512, device=dev)
side_buf = torch.zeros(512, 512, device=dev)
pool = torch.cuda.graph_pool_handle()
items, state = [], {}


def begin():
    g = torch.cuda.CUDAGraph()
    g.capture_begin(pool=pool, capture_error_mode="relaxed")
    state["g"] = g


def cut(action):
    state["g"].capture_end()
    items.append(state["g"])
    items.append(action)
    begin()


def step(hooked):
    w1.grad = w2.grad = None
    h = (xin @ w1).relu()
    if hooked:
        def hook(g):
            print("hook on thread", threading.get_ident(), "current stream", torch.cuda.current_stream().cuda_stream, flush=True)
            # the host action of the replay: an eager in-place op on a static buffer between two segments
            cut(lambda: side_buf.add_(1.0))
            return g
        h.register_hook(hook)
    z = ((h @ w2) ** 2).mean()
    z.backward()
    return z


print("main thread", threading.get_ident(), flush=True)
step(False)
ref1, ref2 = w1.grad.clone(), w2.grad.clone()
s = torch.cuda.Stream(device=dev)
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step(False)
    torch.cuda.synchronize()
    print("capture stream", s.cuda_stream, flush=True)
    begin()
    z = step(True)
    state["g"].capture_end()
    items.append(state["g"])
torch.cuda.synchronize()
print("tape:", [type(i).__name__ for i in items], flush=True)
for rep in range(3):
    xin.mul_(1.0)           # (inputs are static buffers)
    for it in items:
        it.replay() if isinstance(it, torch.cuda.CUDAGraph) else it()
    torch.cuda.synchronize()
    print(f"replay {rep}: |dw1 - ref| {float((w1.grad - ref1).abs().max()):.3e}  |dw2 - ref| {float((w2.grad - ref2).abs().max()):.3e}  "
          f"side_buf {float(side_buf[0, 0])}  loss {float(z):.6f}", flush=True)
print("[segment_probe] OK")
