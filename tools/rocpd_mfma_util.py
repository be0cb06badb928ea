"""Step-level MFMA utilisation (BASELINE.json's "MFMA util %") from ONE rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, no other tracing domain) over `python3 bench.py --workload <w> --no-secondary
--no-cpu-baseline --no-probe`, rocpd sqlite output.

    util = sum over the dispatches of the timed steps of SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x sum GRBM_GUI_ACTIVE / 8)

Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES = cycles the matrix pipe of a SIMD is busy, summed over the 1024 SIMDs;
GRBM_GUI_ACTIVE = cycles the graphics engine is active, summed over the 8 XCDs.  Under --pmc the dispatches of a process run one
at a time, so the denominator is the SERIALISED kernel time of the step in shader-clock cycles: the figure is relative to the clock
the chip actually ran at during each kernel (reported as `effective_clock_ghz`), and it does not see the overlap of the two streams
of an unprofiled step - `util_vs_unprofiled_step` relates the same busy cycles to the unprofiled step time (from the bench record
given with --ms) at that clock.  The timed steps are found by the optimiser's kernel: one `adamw_kernel` dispatch per step, the
last `--steps` of them end the timed steps.

Usage: python tools/rocpd_mfma_util.py <results.db> --steps K [--ms unprofiled_ms_per_step] --workload c2p --out out.md --json out.json"""
import collections
import json
import re
import sqlite3
import sys


def arg(name, default=None):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


db = sqlite3.connect(sys.argv[1])
steps = int(arg("--steps", "3"))
ms = float(arg("--ms", "0") or 0)
wl = arg("--workload", "c2p")
rows = collections.OrderedDict()          # dispatch id -> [kernel, duration ns, mfma busy, gui active]
for name, cn, val, d, disp in db.execute(
        "select kernel_name, counter_name, value, duration, dispatch_id from counters_collection order by dispatch_id"):
    r = rows.setdefault(disp, [name, d, 0.0, 0.0])
    if cn == "SQ_VALU_MFMA_BUSY_CYCLES":
        r[2] += val
    elif cn == "GRBM_GUI_ACTIVE":
        r[3] += val
ids = list(rows)
opt = [i for i in ids if "adamw_kernel" in rows[i][0]]
assert len(opt) > steps, f"{len(opt)} optimiser dispatches: cannot delimit {steps} timed steps"
lo, hi = opt[-steps - 1], opt[-1]          # (the step ends a few launches after AdamW - the plane refresh; those belong to the next window, evenly)
win = [i for i in ids if lo < i <= hi]
busy = sum(rows[i][2] for i in win)
gui = sum(rows[i][3] for i in win)
dur = sum(rows[i][1] for i in win) * 1e-9
cycles = gui / 8.0
util = busy / (1024.0 * cycles)
clock = cycles / dur / 1e9
per = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for i in win:
    n = re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", rows[i][0]))
    n = n if len(n) < 90 else n[:87] + "..."
    a = per[n]
    a[0] += 1; a[1] += rows[i][1] * 1e-9; a[2] += rows[i][2]; a[3] += rows[i][3]
res = {"workload": wl, "steps": steps, "dispatches_per_step": round(len(win) / steps, 1),
       "mfma_busy_cycles_per_step": busy / steps, "serialised_kernel_ms_per_step": 1e3 * dur / steps,
       "effective_clock_ghz": round(clock, 3), "mfma_util_step": round(util, 4),
       "relative_to": "sum of the dispatch durations of one step under --pmc (dispatches serialised), at the shader clock the chip ran at"}
if ms > 0:
    res["unprofiled_ms_per_step"] = ms
    res["util_vs_unprofiled_step"] = round(busy / steps / (1024.0 * ms * 1e-3 * clock * 1e9), 4)
out = [f"# Step-level MFMA utilisation, workload {wl} (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over bench.py)", "",
       f"* timed window: the last {steps} steps ({res['dispatches_per_step']} dispatches per step)",
       f"* sum SQ_VALU_MFMA_BUSY_CYCLES per step = {busy / steps:.4g}; sum GRBM_GUI_ACTIVE / 8 per step = {cycles / steps:.4g} cycles "
       f"= {1e3 * dur / steps:.1f} ms of serialised kernel time at {clock:.2f} GHz",
       f"* **MFMA utilisation of the step = {100 * util:.1f} %** of the matrix pipes' cycles (1024 SIMDs), relative to the serialised "
       "kernel time at the clock the chip ran at"]
if ms > 0:
    out.append(f"* against the UNPROFILED step ({ms:.1f} ms, two streams overlapping) at the same clock: {100 * res['util_vs_unprofiled_step']:.1f} %")
out += ["", "| kernel | dispatches per step | ms per step (profiled) | MFMA busy % of its own time | share of the step's MFMA cycles % |", "|---|---|---|---|---|"]
for n, (c, d, b, g) in sorted(per.items(), key=lambda kv: -kv[1][2])[:14]:
    own = b / (1024.0 * g / 8.0) if g else 0.0
    out.append(f"| `{n}` | {c / steps:.0f} | {1e3 * d / steps:.2f} | {100 * own:.1f} | {100 * b / busy if busy else 0:.1f} |")
txt = "\n".join(out) + "\n"
if arg("--out"):
    open(arg("--out"), "w").write(txt)
if arg("--json"):
    import hashlib, os
    _lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "npvp_amd", "libnpvp_hip.so")
    res["lib_sha256"] = hashlib.sha256(open(_lib, "rb").read()).hexdigest() if os.path.exists(_lib) else None  # bench.py emits the figure only beside THIS build
    json.dump(res, open(arg("--json"), "w"), indent=1)
print(txt)
