"""Per-kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output) as CSV, like `--stats` prints:
name (npvp kernels shortened to their template head), calls, total ms, average us, share of summed kernel time.
Usage: python tools/rocpd_stats.py <results.db> [out.csv] [--by-grid]"""
import re, sqlite3, sys, collections

db = sqlite3.connect(sys.argv[1])
by_grid = "--by-grid" in sys.argv
rows = db.execute("select name, duration, grid_x, workgroup_x, stream_id, start, end from kernels").fetchall()


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*\)$", "", n)
    return n if len(n) < 140 else n[:137] + "..."


acc = collections.OrderedDict()
t0, t1 = min(r[5] for r in rows), max(r[6] for r in rows)
for name, dur, gx, wx, sid, s, e in rows:
    k = short(name) + (f" @{gx // max(wx, 1)}wg" if by_grid else "")
    a = acc.setdefault(k, [0, 0])
    a[0] += 1; a[1] += dur
tot = sum(a[1] for a in acc.values())
out = ["name,calls,total_ms,avg_us,percent"]
for k, (n, d) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    out.append(f"\"{k}\",{n},{d/1e6:.3f},{d/n/1e3:.2f},{100.0*d/tot:.2f}")
out.append(f"\"TOTAL (sum of kernel durations; trace spans {(t1-t0)/1e6:.1f} ms wall)\",{sum(a[0] for a in acc.values())},{tot/1e6:.3f},,100")
txt = "\n".join(out) + "\n"
if len(sys.argv) > 2 and not sys.argv[2].startswith("--"):
    open(sys.argv[2], "w").write(txt)
print("\n".join(out[:45]))
