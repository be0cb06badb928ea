"""Stand-alone reproducer (torch + the HIP runtime only, no npvp_amd kernels) of what round 6 found in the ROCm 7.2 graph replay:
MEMSET NODES of a captured graph are not executed reliably when the runtime replays the graph from prepared AQL packets (its
default).  The FIRST replay after the instantiation is right; from the second on some memset nodes fill part of their buffer (a
quarter of a 60 KB one) with a stale non-zero pattern instead of zeros, or do nothing - with a small graph (CHAINS=8) it is the
4-byte semaphore of a torch reduction that stays uncleared, and the reduction's result is never written.  WHICH nodes break, and
whether any does, depends on unrelated state of the process - what BETWEEN does in front of the replays decides it differently on
different boxes (profiles/r06_graph_alloc_hazard.txt 5f: on one box `none` fails and `memset:1048576` passes, on another the reverse).
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (node-by-node replay) is exact in every variant.  The graph is CHAINS x

    hipMemsetAsync(slot_i, 0)                      <- memset node (what npvp_split_weights_f16 did to its amax table until round 6)
    slot_i = max(slot_i, bound_i)                  <- kernel: raises the slot (the amax producer)
    out_i  = data_i * slot_i[0]                    <- kernel: consumes it (the operand scale)

plus a torch reduction (mean of a large tensor: its 4-byte semaphore is cleared by a memset node too).  After every replay out_i must
equal data_i * bound_i and the mean must be the tensor's mean (slots, outputs and the mean are overwritten with junk before every
replay).  BETWEEN: none | tiny (a 16-float tensor allocated after the capture, one float filled before every replay) |
memset:<bytes> (an eager hipMemsetAsync of another buffer in front of every replay).

    for b in none tiny memset:4 memset:1048576; do BETWEEN=$b VERBOSE=1 python tools/graph_memset_node_repro.py; done            # ROCm default
    for b in none tiny memset:4 memset:1048576; do DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 BETWEEN=$b python tools/graph_memset_node_repro.py; done
"""
import ctypes, os, sys
import torch

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.zeros(1, device=dev)
path = None
with open("/proc/self/maps") as f:
    for line in f:
        if "libamdhip64" in line:
            path = line.split()[-1]
            break
hip = ctypes.CDLL(path)            # (the runtime torch has already loaded: same handle)
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]

CHAINS = int(os.environ.get("CHAINS", 64))
REPLAYS = int(os.environ.get("REPLAYS", 12))
SLOT = int(os.environ.get("SLOT_FLOATS", 15360))         # 61 440 bytes, the size of the training step's amax table
slots = [torch.full((SLOT,), 7.0, device=dev) for _ in range(CHAINS)]
bounds = [torch.full((SLOT,), float(i + 1), device=dev) for i in range(CHAINS)]
data = [torch.rand(1 << 16, device=dev) + 0.5 for _ in range(CHAINS)]
outs = [torch.empty(1 << 16, device=dev) for _ in range(CHAINS)]
big = torch.rand(64 * 28 * 512 * 8, device=dev)
want_mean = float(big.double().mean())


def step():
    s = torch.cuda.current_stream().cuda_stream
    for i in range(CHAINS):
        rc = hip.hipMemsetAsync(slots[i].data_ptr(), 0, SLOT * 4, s)
        assert rc == 0, rc
        torch.maximum(slots[i], bounds[i], out=slots[i])
        torch.mul(data[i], slots[i][0], out=outs[i])
    return big.mean()


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    mean = step()
torch.cuda.synchronize()
between = os.environ.get("BETWEEN", "none")
bad_chain = bad_mean = 0
for r in range(REPLAYS):
    for i in range(CHAINS):
        slots[i].fill_(7.0); outs[i].fill_(-1.0)          # (a replay that skips the memset would still pass; one that clears late fails)
    mean.fill_(-1.0)
    if between == "tiny":
        if r == 0:
            keep = torch.empty(16, device=dev)
        keep[0:1].fill_(1.0)
    elif between.startswith("memset:"):                   # an EAGER hipMemsetAsync of another buffer in front of the replay
        nb = int(float(between.split(":")[1]))
        if r == 0:
            other = torch.empty(max(nb, 4), dtype=torch.uint8, device=dev)
        assert hip.hipMemsetAsync(other.data_ptr(), 0, nb, torch.cuda.current_stream().cuda_stream) == 0
    g.replay()
    torch.cuda.synchronize()
    wrong = [i for i in range(CHAINS) if not torch.equal(outs[i], data[i] * float(i + 1)) or float(slots[i][0]) != float(i + 1)]
    m = float(mean)
    if wrong:
        bad_chain += 1
    if abs(m - want_mean) > 1e-4:
        bad_mean += 1
    if wrong and os.environ.get("VERBOSE") == "1":
        i = wrong[0]
        print(f"  chain {i}: bound {i + 1}, slot[0:4] {[float(v) for v in slots[i][:4]]}, elements of the slot != bound: {int((slots[i] != float(i + 1)).sum())} of {SLOT}, "
              f"out[0] / data[0] = {float(outs[i][0] / data[i][0]):.4f}", flush=True)
    if wrong or abs(m - want_mean) > 1e-4:
        print(f"replay {r}: {len(wrong)} of {CHAINS} chains wrong (first {wrong[:4]}, slot value {float(slots[wrong[0]][0]) if wrong else '-'}), mean {m:.6f} (want {want_mean:.6f})", flush=True)
mode = os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "1 (runtime default)")
print(f"[graph_memset_node_repro] DEBUG_CLR_GRAPH_PACKET_CAPTURE={mode} BETWEEN={between}: {REPLAYS} replays, {bad_chain} with wrong chains, {bad_mean} with a wrong mean -> "
      + ("OK" if bad_chain + bad_mean == 0 else "MEMSET NODES NOT EXECUTED AS CAPTURED"), flush=True)
sys.exit(0 if bad_chain + bad_mean == 0 else 1)
