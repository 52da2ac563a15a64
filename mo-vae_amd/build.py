"""Builds libmovae_hip.so (gfx950) in-tree with hipcc.  No torch headers: the library is a plain
C ABI (include/movae.h) loaded through ctypes, so the ROCm 7.2 toolchain here and the
torch+rocm runtime on the GPU box never have to agree on a C++ ABI."""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libmovae_hip.so")
SOURCES = ["conv_igemm.hip", "bn_act.hip", "eltwise.hip", "losses.hip", "edge.hip", "agg.hip", "vq.hip", "optim.hip", "prior.hip", "api.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build libmovae_hip.so)")


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(BUILD, exist_ok=True)
    hipcc = _hipcc()
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    headers.append(os.path.join(HERE, "..", "include", "movae.h"))
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(BUILD, s + ".o")
        stamp = obj + ".sha"
        dig = _digest([src] + headers)
        objs.append(obj)
        if force or not os.path.exists(obj) or not os.path.exists(stamp) or open(stamp).read() != dig:
            jobs.append((src, obj, stamp, dig))

    def compile_one(job):
        src, obj, stamp, dig = job
        cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        with open(stamp, "w") as f:
            f.write(dig)
        return src

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for done in ex.map(compile_one, jobs):
                if verbose:
                    print("compiled", os.path.basename(done))
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print("linked", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
