"""Per-kernel averages of the counters of one or more rocprofv3 --pmc runs (rocpd sqlite output), as a markdown table.
Usage: python tools/rocpd_pmc.py <results.db> [<results2.db> ...] [--filter substr] [--out file.md]"""
import re, sqlite3, sys, collections

dbs = [a for a in sys.argv[1:] if a.endswith(".db")]
flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else "npvp::gemm"
out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for path in dbs:
    db = sqlite3.connect(path)
    seen = set()
    for name, gs, cn, val, d, disp in db.execute("select kernel_name, grid_size, counter_name, value, duration, dispatch_id from counters_collection"):
        if flt not in name:
            continue
        k = re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", name)) + f" @{gs // 256}wg"
        a = acc[k][cn]; a[0] += 1; a[1] += val
        if (path, disp) not in seen:
            seen.add((path, disp)); dur[k][0] += 1; dur[k][1] += d
names = sorted({c for k in acc for c in acc[k]})
lines = ["| kernel @workgroups | dispatches | avg us (profiled) | " + " | ".join(names) + " |", "|---|---|---|" + "---|" * len(names)]
for k in sorted(acc, key=lambda k: -dur[k][1]):
    n = max(v[0] for v in acc[k].values())
    lines.append(f"| `{k}` | {n} | {dur[k][1] / max(dur[k][0], 1) / 1e3:.1f} | " +
                 " | ".join(f"{acc[k][c][1] / max(acc[k][c][0], 1):.4g}" if c in acc[k] else "-" for c in names) + " |")
txt = "\n".join(lines) + "\n"
if out:
    open(out, "w").write(txt)
print(txt)
