"""Build libhelmnet_hip.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m helmnet_amd.build [--force]

hipcc cross-compiles without a GPU.  The shared object is git-ignored but travels to the GPU
box with the working tree.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libhelmnet_hip.so")
SOURCES = ["hn_api.hip", "hn_spectral.hip", "hn_unet.hip", "hn_mfma.hip", "hn_deep.hip", "hn_deepx.hip", "hn_dcv.hip", "hn_dca.hip", "hn_cs.hip", "hn_train.hip", "hn_rows.hip"]
HEADERS = [os.path.join(CSRC, "hn_internal.h"), os.path.join(CSRC, "hn_vec.h"), os.path.join(CSRC, "hn_dca_pass.inc"), os.path.join(REPO, "include", "helmnet_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-I" + os.path.join(REPO, "include"), "-I" + CSRC,
         "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(obj, [sp] + HEADERS):
            jobs.append([hipcc, *FLAGS, "-c", sp, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
