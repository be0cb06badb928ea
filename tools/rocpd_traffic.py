"""HBM-side traffic per dispatch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE need separate passes: TCC slot
budget, MI355X_MICROARCH.md) of the same command, rocpd sqlite output.  FETCH_SIZE / WRITE_SIZE are reported in KiB; on
gfx950 FETCH_SIZE tallies the 128-B requests of wide (16 B per lane) coalesced reads at 64 B, so it is DOUBLED before it is
compared with a byte count; WRITE_SIZE is exact for 16-B-per-lane stores.
Usage: python tools/rocpd_traffic.py <fetch_results.db> <write_results.db> <out.md> <out.json>"""
import collections, json, re, sqlite3, sys


def load(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    db = sqlite3.connect(path)
    for name, val in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        n = re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", name))
        full = n if len(n) < 100 else n[:97] + "..."
        base = re.sub(r"<.*$", "", n)
        m = re.match(r"(npvp::gemm_(?:wide|f16)_kernel<\d+, \d+, \d+, \d+), .*>", n)
        if m:               # the tile instantiations of the wide kernel are different kernels to bench.py (npvp_gemm_kernel_id 2 / 4)
            base = m.group(1) + ">"
        for k in (full, "POOL:" + base):
            a = acc[k]; a[0] += 1; a[1] += val * 1024.0
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = ["# HBM-side traffic per dispatch (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes over `python bench.py`)", "",
       "FETCH_SIZE is shown raw and x2 (gfx950 tallies 128-B requests of 16 B/lane reads at 64 B, MI355X_MICROARCH.md);",
       "WRITE_SIZE is exact.  npvp kernels with >= 8 dispatches.", "",
       "| kernel | dispatches | FETCH raw MB | FETCH x2 MB | WRITE MB | FETCH x2 + WRITE MB |", "|---|---|---|---|---|---|"]
js = {"per_instantiation": {}, "pooled": {}}
for k, (n, tot) in sorted(fetch.items(), key=lambda kv: -kv[1][1]):
    if not (k.startswith("npvp::") or k.startswith("POOL:npvp::")) or n < 8 or k not in write:
        continue
    f, w = tot / n, write[k][1] / write[k][0]
    if k.startswith("POOL:"):
        js["pooled"][k[5:]] = {"dispatches": n, "fetch_x2_bytes": 2 * f, "write_bytes": w, "hbm_bytes_per_dispatch": 2 * f + w}
    else:
        out.append(f"| `{k}` | {n} | {f/1e6:.1f} | {2*f/1e6:.1f} | {w/1e6:.1f} | {(2*f+w)/1e6:.1f} |")
        js["per_instantiation"][k] = {"dispatches": n, "fetch_raw_bytes": f, "fetch_x2_bytes": 2 * f, "write_bytes": w}
out += ["", "## pooled over template instantiations (what `roofline.traffic` of bench.py reports)", "",
        "| kernel | dispatches | FETCH x2 + WRITE, MB per dispatch |", "|---|---|---|"]
for k, v in sorted(js["pooled"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_dispatch"] * kv[1]["dispatches"]):
    out.append(f"| `{k}` | {v['dispatches']} | {v['hbm_bytes_per_dispatch']/1e6:.1f} |")
open(sys.argv[3], "w").write("\n".join(out) + "\n")
import hashlib, os
_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "npvp_amd", "libnpvp_hip.so")
js["lib_sha256"] = hashlib.sha256(open(_lib, "rb").read()).hexdigest() if os.path.exists(_lib) else None      # bench.py emits these figures only beside THIS build
json.dump(js, open(sys.argv[4], "w"), indent=1)
print("\n".join(out[:40]))
