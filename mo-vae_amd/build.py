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


def _stale(force):
    """(objects, compile jobs) against the digests on disk."""
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
    return objs, jobs


def build(force=False, verbose=True):
    """Rebuilds what changed.  Safe under concurrent callers (every rank of a torchrun launch imports the package): the
    rebuild is serialised by an exclusive flock on _build/.lock, the digests are re-checked after the lock is taken (a peer
    may have finished the work meanwhile), and objects / stamps / the .so are written to temporary names and os.replace()d
    into place, so a process that dlopens the library concurrently sees the old file or the new one, never a partial one."""
    import fcntl

    os.makedirs(BUILD, exist_ok=True)
    hipcc = _hipcc()
    objs, jobs = _stale(force)
    if not jobs and os.path.exists(LIB):
        return LIB  # (the common case takes no lock)
    with open(os.path.join(BUILD, ".lock"), "w") as lockf:
        fcntl.flock(lockf, fcntl.LOCK_EX)
        try:
            objs, jobs = _stale(force)
            tag = f".tmp{os.getpid()}"

            def compile_one(job):
                src, obj, stamp, dig = job
                cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", src, "-o", obj + tag]
                r = subprocess.run(cmd, capture_output=True, text=True)
                if r.returncode != 0:
                    if os.path.exists(obj + tag):
                        os.remove(obj + tag)
                    raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
                if verbose and r.stderr.strip():
                    print(r.stderr, file=sys.stderr)
                os.replace(obj + tag, obj)
                with open(stamp + tag, "w") as f:
                    f.write(dig)
                os.replace(stamp + tag, stamp)
                return src

            if jobs:
                with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
                    for done in ex.map(compile_one, jobs):
                        if verbose:
                            print("compiled", os.path.basename(done))
            if jobs or not os.path.exists(LIB):
                cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + tag] + objs
                r = subprocess.run(cmd, capture_output=True, text=True)
                if r.returncode != 0:
                    if os.path.exists(LIB + tag):
                        os.remove(LIB + tag)
                    raise RuntimeError(f"link failed:\n{r.stderr}")
                os.replace(LIB + tag, LIB)
                if verbose:
                    print("linked", LIB)
        finally:
            fcntl.flock(lockf, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
