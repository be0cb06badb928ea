"""Build libnpvp_hip.so (gfx950) in-tree from npvp_amd/csrc/*.hip with hipcc.

The shared object lives next to this file (npvp_amd/libnpvp_hip.so): it is git-ignored but
travels to the GPU box with the repo snapshot.  Incremental: a source is recompiled only when it
(or a header of csrc/) is newer than its object file.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libnpvp_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hipcc = _hipcc()
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + ".o")
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        # NPVP_HIPCC_EXTRA: extra flags for one-off builds on the GPU box (e.g. -save-temps, -Rpass-analysis=kernel-resource-usage); never set
        # by the product build
        cmd = [hipcc] + FLAGS + os.environ.get("NPVP_HIPCC_EXTRA", "").split() + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose:
            print(f"[npvp_amd.build] compiled {os.path.basename(src)}", file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in srcs]
    stale = [f for f in os.listdir(OBJ) if f.endswith(".o") and os.path.join(OBJ, f) not in objs]
    for f in stale:                      # a source file was removed: its object must not linger in the library
        os.remove(os.path.join(OBJ, f))
    if force or jobs or stale or not os.path.exists(LIB) or any(_newer(o, LIB) for o in objs):
        # --no-undefined: a symbol dropped from one translation unit must fail HERE, not at first call on the GPU box
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-Wl,--no-undefined", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[npvp_amd.build] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
