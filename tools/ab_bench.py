"""A/B of one module attribute on a bench workload, both settings in ONE process on one box (boxes differ by 2 - 3 %):
    python tools/ab_bench.py <workload> <module.attr> <value A> <value B> [steps]
e.g. python tools/ab_bench.py c2 ops.DROP_PATH_IN_GEMM_ROWS 32768 1000000000 [--set=ops.FusedLinearBwd.enabled=False ...]
Runs A, B, A, B (each: build, 3 warm-up steps, `steps` timed steps) and prints ms per step."""
import argparse
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch          # noqa: E402
import bench          # noqa: E402
from npvp_amd import ops, sched, dp          # noqa: E402

def resolve(target):
    modname, attr = target.rsplit(".", 1)
    obj = {"ops": ops, "sched": sched}[modname.split(".")[0]]
    for part in modname.split(".")[1:]:
        obj = getattr(obj, part)
    return obj, attr


fixed = [a[6:] for a in sys.argv[1:] if a.startswith("--set=")]          # --set=ops.X.attr=value : held for the whole run
argv = [a for a in sys.argv[1:] if not a.startswith("--set=")]
for spec in fixed:
    t, v = spec.split("=", 1)
    o, a = resolve(t)
    setattr(o, a, ast.literal_eval(v))
wl, target, va, vb = argv[0:4]
steps = int(argv[4]) if len(argv) > 4 else 8
obj, attr = resolve(target)
args = argparse.Namespace(flavour="predictor", graph=False, probe_all=False, graph_streams=1, dp_fused_trial="auto")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
res = {va: [], vb: []}
for rep in range(2):
    for v in (va, vb):
        setattr(obj, attr, ast.literal_eval(v))
        r = bench.run_workload(wl, steps, 3, args, 0, 1, dev, probe=False)
        res[v].append(round(r["ms"], 2))
        print(f"{target} = {v}: {r['ms']:.2f} ms/step", flush=True)
print({k: v for k, v in res.items()})
