"""Build libuavenv.so (hand-written HIP for gfx950) in-tree with hipcc.

The .so is git-ignored but travels to the GPU box with the repo snapshot.  hipcc
cross-compiles without a GPU, so this also is the CPU-side "does it build" check.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libuavenv.so")
SOURCES = ["uavenv.hip", "replay.hip", "learner.hip", "rrt.hip", "per.hip", "loop.hip", "p2p.hip", "coll.hip", "sac.hip", "fed.hip"]
HEADERS = ["uavenv_device.hpp", "qnet_device.hpp", os.path.join("..", "..", "include", "uavenv.h")]
# -ffp-contract=off: the reward / collision arithmetic must round like the reference's
# separate multiplies and adds (no FMA fusion); see csrc/uavenv_device.hpp.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-unused-value", "-ldl"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libuavenv.so cannot be built (there is no CPU fallback)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile under an exclusive file lock into a temporary name, then rename: `torchrun bench.py --gpus N` starts N
    ranks that may all find the library stale; one builds, the others wait on the lock and find it fresh."""
    if not force and not needs_build():
        return LIB_PATH
    import fcntl
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():       # another process built it while this one waited
                return LIB_PATH
            return _build_locked(verbose, force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose: bool, force: bool = False) -> str:
    """One object per translation unit (compiled in parallel, reused while neither the source nor a header changed),
    then one link.  Objects live in build/ (git-ignored, not shipped)."""
    from concurrent.futures import ThreadPoolExecutor
    extra = ["-DUAVENV_PHASE_PROFILE"] if os.environ.get("UAVENV_PHASE_PROFILE") else []   # diagnostics build
    extra += os.environ.get("UAVENV_EXTRA_FLAGS", "").split()                              # A/B experiments
    hipcc = _hipcc()
    cflags = [f for f in FLAGS if f not in ("-shared", "-ldl")] + extra
    objdir = os.path.join(PKG_DIR, "build", "obj" + ("_" + "_".join(extra).replace("-", "").replace("=", "") if extra else ""))
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.normpath(os.path.join(CSRC, h))) for h in HEADERS)
    hdr_t = max(hdr_t, os.path.getmtime(os.path.abspath(__file__)))

    def compile_one(src):
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_t):
            return obj, None
        cmd = [hipcc] + cflags + ["-c", path, "-o", obj + ".tmp.%d" % os.getpid()]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            return obj, "hipcc failed on %s:\n%s%s" % (src, res.stdout, res.stderr)
        os.replace(obj + ".tmp.%d" % os.getpid(), obj)
        return obj, None

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        results = list(ex.map(compile_one, SOURCES))
    errs = [e for _, e in results if e]
    if errs:
        raise RuntimeError("\n".join(errs))
    tmp = "%s.tmp.%d" % (LIB_PATH, os.getpid())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [o for o, _ in results] + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)                          # atomic: a concurrent CDLL never maps a half-written file
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
