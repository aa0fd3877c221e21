"""Build recipe for libkmx.so (hipcc, gfx950 only).  `python -m kmcex_amd.build` or __graft_entry__.build()."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libkmx.so")
SOURCES = ["kernels.hip", "rest_device.hip", "kmx_api.hip", "kmc_reader.cpp", "strpack.cpp"]
HEADERS = ["device_common.h", "kmx_types.h", "kmc_reader.h", "strpack.h", "range_kernels.h", "multi_build.h", "range_host.h", os.path.join("..", "..", "include", "kmx.h")]


def hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libkmx.so cannot be built (there is no CPU fallback)")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    objs, cmds = [], []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
               "-c", os.path.join(CSRC, src), "-o", obj]
        if src.endswith(".cpp"):
            cmd.insert(1, "-x")
            cmd.insert(2, "c++")
        if verbose:
            print(" ".join(cmd), flush=True)
        objs.append(obj)
        cmds.append(cmd)
    procs = [subprocess.Popen(c) for c in cmds]                  # the translation units compile side by side
    failed = [c for c, p in zip(cmds, procs) if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
