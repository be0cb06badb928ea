"""Where the HOST time of a small-shard training step goes (the 8-clip c3 / c4 shards are bound by launch count).
    python tools/host_profile.py [workload key, default c4] [steps]
Prints (1) cProfile of the step function by own time, (2) torch.profiler's CPU-op table, (3) the python call sites (innermost
npvp_amd frame) of the aten ops that launch small kernels: copy_, add, fill_, zero_, mul ..."""
import cProfile
import collections
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import npvp_amd  # noqa: E402
from npvp_amd import ops  # noqa: E402
from npvp_amd.trainer import load_config  # noqa: E402


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "c4"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda:0")
    cfg_file, name, B, To, Tp = bench.WORKLOADS[key]
    cfg = load_config(os.path.join(ROOT, "configs", cfg_file), B, To, Tp)
    P = cfg["Predictor"]
    torch.manual_seed(0)
    model = npvp_amd.build_predictor_from_cfg(npvp_amd.Predictor, P, To, Tp).to(dev)
    model.train()
    opt = npvp_amd.FlatAdamW(model, lr=P["predictor_lr"], clip_module=model.transformer, max_grad_norm=P["max_grad_norm"])
    ops.rng.manual_seed(1, dev)
    past = torch.relu(torch.randn(B, To, 512, 8, 8) * 0.1 + 0.05).to(dev)
    fut = torch.relu(torch.randn(B, Tp, 512, 8, 8) * 0.1 + 0.05).to(dev)

    def step():
        return npvp_amd.predictor_train_step(model, opt, past, fut, P["lam_PF_L1"], P["KL_beta"], P["max_grad_norm"], sync=False)

    torch.autograd.set_multithreading_enabled(False)        # backward on this thread: cProfile and the dispatch mode see it
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"[{key}] {B} clips: host {1000 * th / steps:.2f} ms/step, wall {1000 * (time.perf_counter() - t0) / steps:.2f} ms/step")

    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(70)
    print(s.getvalue()[:14000])

    # aten ops that launch small kernels, by python call site (dispatch mode; backward runs on this thread)
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    sites = collections.Counter()
    names = collections.Counter()
    shapes = collections.Counter()

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            n = str(func)
            names[n] += 1
            if "add" in n and args and hasattr(args[0], "shape"):
                shapes[(n, tuple(args[0].shape), tuple(args[1].shape) if len(args) > 1 and hasattr(args[1], "shape") else None)] += 1
            if any(k in n for k in ("copy_", "add", "fill_", "zero_", "mul", "clone", "sum", "div", "sub", "cat", "zeros", "ones",
                                    "select", "_to_copy", "neg", "mean", "sqrt", "exp", "where", "stack", "index")):
                fr = [f for f in traceback.extract_stack(limit=14) if "npvp_amd" in f.filename]
                site = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].name}" if fr else "?"
                sites[(n, site)] += 1
            return func(*args, **(kwargs or {}))

    with Log():
        for _ in range(steps):
            step()
    torch.cuda.synchronize()
    for n, c in names.most_common(40):
        print(f"{c / steps:8.1f}/step  {n}")
    print()
    for k, c in shapes.most_common(40):
        print(f"{c / steps:8.1f}/step  {k}")
    print()
    for (n, site), c in sites.most_common(90):
        print(f"{c / steps:8.1f}/step  {n:34s} {site}")


if __name__ == "__main__":
    main()
