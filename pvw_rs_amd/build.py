"""Builds the HIP libraries (hipcc, gfx950 only) in-tree:

  pvw_rs_amd/libpvw_hip.so          the shipped library: shape-selected kernel schedules, no environment lookups
  pvw_rs_amd/libpvw_hip_tuning.so   the measurement build (-DPVW_TUNING=1, include/pvw_hip_tuning.h): schedule
                                    selectors, timing ablations and the bandwidth probe for tools/*.sh, bench.py's
                                    read_probe leg and tests/test_gpu_tuning.py

hipcc cross-compiles without a GPU; the .so files travel to the GPU box with the snapshot
(they are git-ignored, not gpurun-ignored)."""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(HERE, "..", "include")
LIB = os.path.join(HERE, "libpvw_hip.so")
LIB_TUNING = os.path.join(HERE, "libpvw_hip_tuning.so")
SOURCES = ["pvw_mac.hip", "pvw_poly.hip", "pvw_decrypt.hip", "pvw_decode_kernels.hip", "pvw_gemm.hip", "pvw_capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _deps():
    """every source and header either library is built from (globbed, so a new header cannot be forgotten)"""
    return ([os.path.join(CSRC, s) for s in SOURCES] + sorted(glob.glob(os.path.join(CSRC, "*.h")))
            + sorted(glob.glob(os.path.join(INCLUDE, "*.h"))) + [os.path.abspath(__file__)])


def _stale(lib: str) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps())


def _build_one(lib: str, extra, tag: str, verbose: bool):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "..", "build", "obj" + tag)
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for s in SOURCES:
        o = os.path.join(objdir, s + ".o")
        cmd = [hipcc] + FLAGS + list(extra) + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(o)
    return objs, procs


def build(force: bool = False, verbose: bool = False, tuning: bool = True, shipped: bool = True) -> str:
    """Builds whatever is stale (both libraries by default; the four hipcc compiles run side by side)."""
    jobs = []
    if shipped and (force or _stale(LIB)):
        jobs.append((LIB,) + _build_one(LIB, [], "", verbose))
    if tuning and (force or _stale(LIB_TUNING)):
        extra = ["-DPVW_TUNING=1"]
        if os.environ.get("PVW_GEMM_ABLATE"):          # compile-time ablation bits of the digit GEMM (tuning build only)
            extra.append("-DPVW_GEMM_ABLATE=" + os.environ["PVW_GEMM_ABLATE"])
        if os.environ.get("PVW_GEMM_RPW"):             # row tiles per wave of the digit GEMM (experiment)
            extra.append("-DPVW_GEMM_RPW=" + os.environ["PVW_GEMM_RPW"])
        if os.environ.get("PVW_PACKED_WPC"):           # workgroups per CU the packed mac_rows is register-allocated for (experiment)
            extra.append("-DPVW_PACKED_WPC=" + os.environ["PVW_PACKED_WPC"])
        jobs.append((LIB_TUNING,) + _build_one(LIB_TUNING, extra, "_tuning", verbose))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for lib, objs, procs in jobs:
        for p, cmd in procs:
            if p.wait() != 0:
                raise RuntimeError("hipcc failed: " + " ".join(cmd))
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--quiet" not in sys.argv, shipped="--tuning-only" not in sys.argv))
