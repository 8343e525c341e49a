"""Builds libm355seg.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libm355seg.so")
SOURCES = ["abi.cpp", "conv3d.hip", "conv3d_f32x3.hip", "conv3d_h16.hip", "act16.hip", "train16.hip", "norm.hip", "elementwise.hip", "convt.hip", "loss_patch_eval.hip", "ensemble.hip",
           "blur_weights.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("common.hpp", "h16.hpp", "conv3d_common.hpp", "h16_epilogue.hpp")] + \
    [os.path.join(HERE, "..", "include", "m355seg.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=True, stamps=False):
    """Compile every HIP source to an object (in parallel) and link the shared library.
    stamps=True: the diagnostic build libm355seg_dbg.so (-DM355_H16_STAMPS: in-kernel cycle stamps of the 16-bit
    conv kernel, tools/h16_stamps.py; loaded instead of the product library through M355_LIB_PATH)."""
    objdir = os.path.join(HERE, "build_dbg" if stamps else "build")
    lib = os.path.join(HERE, "libm355seg_dbg.so") if stamps else LIB
    flags = FLAGS + (["-DM355_H16_STAMPS"] if stamps else [])
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + HEADERS):
            cmd = [hipcc] + flags + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print("[m355seg build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, stamps="--stamps" in sys.argv))
