"""Counter values per dispatch IN LAUNCH ORDER from rocprofv3 --pmc runs (rocpd sqlite): python tools/rocpd_seq.py <db> [<db2>] [--filter s]"""
import re, sqlite3, sys, collections
dbs = [a for a in sys.argv[1:] if a.endswith(".db")]
flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else "npvp::gemm"
rows = collections.OrderedDict()
for path in dbs:
    db = sqlite3.connect(path)
    for disp, name, gs, cn, val in db.execute("select dispatch_id, kernel_name, grid_size, counter_name, value from counters_collection order by dispatch_id"):
        if flt in name:
            rows.setdefault(disp, {"name": re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", name))[:60], "wg": gs // 256})[cn] = val
for d, r in rows.items():
    print(d, r["name"], f"@{r['wg']}wg", " ".join(f"{k}={v * 1024 / 1e6:.1f}MB" for k, v in r.items() if k not in ("name", "wg")))
