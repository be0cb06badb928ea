"""What the compiler made of every kernel: per kernel the instruction count by class (scalar / vector / MFMA), integer-division
sequences (v_rcp_iflag_f32 = one 32-bit division, 64-bit ones show as mul_hi chains), VGPRs, spills and scratch bytes, plus the
loops the optimizer refused to unroll (-Rpass-missed=loop-unroll: an epilogue that indexes its accumulator tiles in such a loop keeps
them in scratch memory).  Runs hipcc --cuda-device-only -S on npvp_amd/csrc/*.hip; no GPU needed.
Usage: python tools/isa_audit.py [file.hip ...] [--min N]      (kernels with fewer than N instructions are left out, default 50)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "npvp_amd", "csrc")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
minimum = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 50
if "--min" in sys.argv:
    args = [a for a in args if a != sys.argv[sys.argv.index("--min") + 1]]
files = args or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def demangle(names):
    if not names:
        return []
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.split("\n")
    return [re.sub(r"\(.*", "", o) for o in out]


print(f"{'file':14s} {'instr':>6s} {'scalar':>6s} {'vector':>6s} {'mfma':>5s} {'idiv':>4s} {'mulhi':>5s} {'vgpr':>4s} {'spill':>5s} {'scratch':>7s}  kernel")
for f in files:
    src = os.path.join(CSRC, os.path.basename(f))
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}", "-S",
                            "--cuda-device-only", "-Rpass-missed=loop-unroll", "-o", asm, src], capture_output=True, text=True)
        if r.returncode:
            print(f"{f}: hipcc failed\n{r.stderr[-2000:]}")
            continue
        missed = sorted(set(re.findall(r"(\S+\.(?:hip|h):\d+):\d+: remark: Unable to (?:fully )?unroll", r.stderr)))
        lines = open(asm).read().split("\n")
    cur, stats, meta = None, {}, {}
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            stats[cur] = dict(n=0, s=0, v=0, mfma=0, idiv=0, mulhi=0)
            continue
        if ln.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"\s+\.name:\s+(\S+)", ln)
        if m:
            metak = m.group(1)
            meta[metak] = {}
        m = re.match(r"\s+\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size):\s+(\d+)", ln)
        if m and meta:
            meta[metak][m.group(1)] = int(m.group(2))
        if cur is None:
            continue
        t = ln.strip().split(" ")[0].split("\t")[0]
        if not t or t[0] in ";.":
            continue
        st = stats[cur]
        st["n"] += 1
        st["s"] += t.startswith("s_")
        st["v"] += t.startswith("v_")
        st["mfma"] += t.startswith("v_mfma")
        st["idiv"] += t.startswith("v_rcp_iflag")
        st["mulhi"] += t in ("s_mul_hi_u32", "v_mul_hi_u32")
    keys = [k for k, st in stats.items() if st["n"] >= minimum]
    for k, name in zip(keys, demangle(keys)):
        st, mt = stats[k], meta.get(k, {})
        print(f"{os.path.basename(f):14s} {st['n']:6d} {st['s']:6d} {st['v']:6d} {st['mfma']:5d} {st['idiv']:4d} {st['mulhi']:5d} "
              f"{mt.get('vgpr_count', 0):4d} {mt.get('vgpr_spill_count', 0):5d} {mt.get('private_segment_fixed_size', 0):7d}  {name[:110]}")
    for m in missed:
        print(f"{os.path.basename(f):14s} LOOP NOT UNROLLED as the pragma asks: {m}")
