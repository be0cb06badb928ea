"""Kernel launch ORDER of a rocprofv3 --kernel-trace run (rocpd sqlite), compact: python tools/rocpd_order.py <db> [first] [count]
prints `count` consecutive dispatches starting at index `first` (default: the last 400) as  stream  name  duration_us"""
import re
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel_dispatch" in t][0]
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
name_c = "name" if "name" in cols else "kernel_name"
rows = list(db.execute(f"select {name_c}, start, end, stream_id from {view} order by start")) if "stream_id" in cols else \
    [r + (0,) for r in db.execute(f"select {name_c}, start, end from {view} order by start")]
first = int(sys.argv[2]) if len(sys.argv) > 2 else max(0, len(rows) - 400)
count = int(sys.argv[3]) if len(sys.argv) > 3 else 400
for n, s, e, st in rows[first:first + count]:
    n = re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", n)).replace("npvp::", "").replace("at::native::", "at:")[:70]
    print(f"{st:>4} {n:70s} {(e - s) / 1e3:8.1f}")
print(len(rows), "dispatches; columns:", cols)
