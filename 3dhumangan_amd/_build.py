"""Build libh3d.so (hand-written HIP for gfx950) in-tree with hipcc.  No GPU needed to compile."""
import concurrent.futures as cf
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libh3d.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-inline-asm"]

# Per-file additions.  field_x3.hip: without LLVM's post-register-allocation scheduler the hand-placed section order of the render
# engine survives as written -- same-lease A/B of bench.py, two leases: render stage 16.30 / 16.17 -> 15.79 / 15.98 ms and 16.36 /
# 16.31 -> 16.22 / 16.13 ms (profiles/r5_ab_sched_flags.json).  The same flag costs synthesis_x3.hip 1.2-1.6 %, so it stays per file.
FILE_FLAGS = {"field_x3.hip": ["-mllvm", "-enable-post-misched=false"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build libh3d.so)")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(obj, deps):
    return not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps)


def build_lib(force=False, verbose=False):
    """Compile every csrc/*.hip to an object (parallel), link libh3d.so.  Returns the library path."""
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "h3d.h"))
    objs, jobs = [], []
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        if force or _stale(obj, [src, os.path.abspath(__file__)] + headers):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {os.path.basename(src)}:\n{r.stderr[-4000:]}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    if jobs or force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return LIB


if __name__ == "__main__":
    import sys
    print(build_lib(force="--force" in sys.argv, verbose=True))
