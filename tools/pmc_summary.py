"""Summarise rocprofv3 --pmc passes of bench.py into profiles/: per kernel (short name, grid size) the average counter
value per dispatch.  HBM traffic per MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE come from SEPARATE passes (TCC slot
budget), are reported in KiB, and on gfx950 FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) coalesced reads
at 64 B - it is doubled before comparing with a byte count; WRITE_SIZE is exact for 16-B-per-lane stores.
Usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.md> <out.json>"""
import csv, json, re, sys, collections

csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(npvp::[\w]+(?:<[^>]*>)?)", name)
    if m:
        return m.group(1)
    return name[:60]


def load(path):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        key = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])), r["Counter_Name"])
        a = acc[key]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc


fetch, write = load(sys.argv[1]), load(sys.argv[2])
rows = {}
for (k, wg, c), (n, tot) in list(fetch.items()) + list(write.items()):
    rows.setdefault((k, wg), {})[c] = (n, tot / n * 1024.0)
out = ["# HBM-side traffic per dispatch (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes over `bench.py`)", "",
       "FETCH_SIZE is shown raw and x2 (gfx950 tallies 128-B requests of 16 B/lane reads at 64 B, MI355X_MICROARCH.md);",
       "WRITE_SIZE is exact.  Only npvp kernels with >= 8 dispatches are listed.", "",
       "| kernel | workgroups | dispatches | FETCH raw MB | FETCH x2 MB | WRITE MB |", "|---|---|---|---|---|---|"]
js = {}
for (k, wg), d in sorted(rows.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", (0, 0))[1] * kv[1].get("FETCH_SIZE", (0, 0))[0])):
    if not k.startswith("npvp::") or "FETCH_SIZE" not in d or "WRITE_SIZE" not in d or d["FETCH_SIZE"][0] < 8:
        continue
    f, w = d["FETCH_SIZE"][1], d["WRITE_SIZE"][1]
    out.append(f"| `{k}` | {wg} | {d['FETCH_SIZE'][0]} | {f/1e6:.1f} | {2*f/1e6:.1f} | {w/1e6:.1f} |")
    js[f"{k}@{wg}"] = {"dispatches": d["FETCH_SIZE"][0], "fetch_raw_bytes": f, "fetch_x2_bytes": 2 * f, "write_bytes": w}
pool = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
for (k, wg, c), (n, tot) in fetch.items():
    pool[k][0] += n; pool[k][1] += tot * 1024.0
for (k, wg, c), (n, tot) in write.items():
    pool[k][2] += n; pool[k][3] += tot * 1024.0
out += ["", "## pooled over all grid sizes (what `roofline.traffic` of bench.py reports for the dominant kernel)", "",
        "| kernel | dispatches | FETCH x2 + WRITE, MB per dispatch |", "|---|---|---|"]
js["pooled"] = {}
for k, (nf, f, nw, w) in sorted(pool.items(), key=lambda kv: -kv[1][1]):
    if k.startswith("npvp::") and nf >= 8 and nw >= 8:
        b = 2 * f / nf + w / nw
        out.append(f"| `{k}` | {nf} | {b/1e6:.1f} |")
        js["pooled"][k] = {"dispatches": nf, "hbm_bytes_per_dispatch": b}
open(sys.argv[3], "w").write("\n".join(out) + "\n")
json.dump(js, open(sys.argv[4], "w"), indent=1)
print("\n".join(out[:30]))
