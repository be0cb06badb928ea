"""Does a kernel that re-reads a tensor the previous kernel just wrote run faster than from HBM (256 MB Infinity Cache)?
write X MB (fill), then read it (sum): the read's time and GB/s by size."""
import torch
dev = "cuda:0"
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev)
    y = torch.empty(n, device=dev)
    res = []
    for mode in ("write->read", "read cold"):
        ts = []
        for it in range(6):
            if mode == "write->read":
                x.fill_(1.0)
            else:
                y.fill_(1.0)          # evicts x (for sizes above the cache) / a fair amount of it
                y.fill_(2.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            s = x.sum()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[len(ts) // 2]
        res.append(f"{mode}: {t * 1e3:8.1f} us {mb / 1024 / (t * 1e-3):7.0f} GB/s")
    print(f"{mb:5d} MB  " + "   ".join(res), flush=True)
