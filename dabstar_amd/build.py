"""Builds libdabx.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

Every source of csrc/ is compiled to its own object (in parallel, rebuilt only when it or a header is newer) and the objects
are linked into one fat shared library: `hipcc --offload-arch=gfx950 -c` per file, then `hipcc -shared`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
OUT = os.path.join(HERE, "libdabx.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
         "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "dabx.h")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in sources() + headers())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in headers())
    jobs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([HIPCC] + FLAGS + ["-x", "hip", "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s) + ".o") for s in sources()]
    run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT])
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print("built", OUT)
