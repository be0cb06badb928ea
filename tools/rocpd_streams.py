"""Per-STREAM view of a rocprofv3 --kernel-trace run (rocpd sqlite output): for the last `steps` steps of a benchmark trace (the window
is cut at the `steps + 1` last launches of a marker kernel that runs once per step - npvp::adamw_kernel), per stream: launches,
summed kernel time, busy time (union of intervals), idle time inside the window and how it splits into gaps by size, and the
kernels that lead its time.  The 8-clip shards are bound by kernel COUNT: this shows how much of a step is gaps between kernels.
Usage: python tools/rocpd_streams.py <results.db> [steps] [out.md]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall()


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*\)$", "", n)
    return n if len(n) < 90 else n[:87] + "..."


marks = [r[1] for r in rows if "adamw_kernel" in r[0]]
if len(marks) > steps:
    t0, t1 = marks[-steps - 1], marks[-1]
else:
    t0, t1, steps = rows[0][1], rows[-1][2], max(1, len(marks))
win = [r for r in rows if t0 <= r[1] < t1]
span = (t1 - t0) / 1e6
out = [f"# window: the last {steps} steps, {span / steps:.2f} ms per step, {len(win) / steps:.0f} launches per step", ""]
by_stream = collections.defaultdict(list)
for r in win:
    by_stream[r[3]].append(r)
# union over all streams
ev = sorted((r[1], r[2]) for r in win)
busy_all, cur_s, cur_e = 0, None, None
for s, e in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy_all += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy_all += (cur_e - cur_s) if cur_e is not None else 0
out.append(f"device busy (union over streams): {busy_all / 1e6 / steps:.2f} ms per step = {100.0 * busy_all / (t1 - t0):.1f} % of the window")
out.append("")
for sid, rs in sorted(by_stream.items(), key=lambda kv: -sum(r[2] - r[1] for r in kv[1])):
    tot = sum(r[2] - r[1] for r in rs)
    gaps = [rs[i + 1][1] - rs[i][2] for i in range(len(rs) - 1)]
    pos = [g for g in gaps if g > 0]
    bins = [(0, 2e3), (2e3, 5e3), (5e3, 1e4), (1e4, 3e4), (3e4, 1e5), (1e5, 1e12)]
    hist = ", ".join(f"{lo / 1e3:.0f}-{'inf' if hi > 1e11 else f'{hi / 1e3:.0f}'} us: {sum(1 for g in pos if lo <= g < hi) / steps:.0f} ({sum(g for g in pos if lo <= g < hi) / 1e6 / steps:.2f} ms)"
                     for lo, hi in bins)
    out.append(f"## stream {sid}: {len(rs) / steps:.0f} launches per step, kernel time {tot / 1e6 / steps:.2f} ms per step, "
               f"gaps {sum(pos) / 1e6 / steps:.2f} ms per step (overlapping launches: {sum(1 for g in gaps if g <= 0) / steps:.0f})")
    out.append(f"gaps by size, per step: {hist}")
    acc = collections.OrderedDict()
    for n, s, e, _ in rs:
        a = acc.setdefault(short(n), [0, 0])
        a[0] += 1; a[1] += e - s
    out.append("| kernel | launches per step | ms per step | average us |")
    out.append("|---|---|---|---|")
    for k, (n, d) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:28]:
        out.append(f"| `{k}` | {n / steps:.1f} | {d / 1e6 / steps:.3f} | {d / n / 1e3:.1f} |")
    out.append("")
txt = "\n".join(out) + "\n"
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(txt)
print(txt[:6000])
