"""Builds pvw_rs_amd/libpvw_hip.so (hipcc, gfx950 only) in-tree.

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the snapshot
(it is git-ignored, not gpurun-ignored)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpvw_hip.so")
SOURCES = ["pvw_kernels.hip", "pvw_capi.hip"]
HEADERS = ["pvw_arith.h", "pvw_bignum.h", "pvw_chacha.h", "pvw_kernels.h",
           os.path.join("..", "..", "include", "pvw_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    os.makedirs(os.path.join(HERE, "..", "build", "obj"), exist_ok=True)
    procs = []
    for s in SOURCES:
        o = os.path.join(HERE, "..", "build", "obj", s + ".o")
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(o)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
