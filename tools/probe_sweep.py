#!/usr/bin/env python3
"""Delivered read bandwidth against bytes in flight (tuning build; run on the GPU box):
    python tools/probe_sweep.py [config]
Sweeps the read probe over tiles in flight per wave (U, 2U when double-buffered) and workgroups resident per CU
(capped with unused LDS) on the resident public key of `config` -- the access pattern of mac_rows without its
arithmetic, r-hat staging or epilogue."""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import pvw_rs_amd as P  # noqa: E402
from pvw_rs_amd import _ffi  # noqa: E402

_ffi.select("tuning")   # the probes live in the measurement build
from pvw_rs_amd import workloads as W  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
n, k, l, L, _ = W.ENCRYPT_CONFIGS[cfg]
p = (P.PvwParametersBuilder().set_parties(n).set_dimension(k).set_l(l).set_moduli(W.bench_moduli(L)).build())
gpk = P.GlobalPublicKey.new(P.PvwCrs.new_deterministic(p, W.SEED_A))
gpk.fill_uniform(W.SEED_B)


def run(u, dbuf, wgs):
    lds = 0 if wgs == 0 else (160 * 1024 // wgs) & ~255
    sec, nb = C.c_double(0.0), C.c_uint64(0)
    p._call("pvw_tuning_read_probe", 20, u, dbuf, lds, C.byref(sec), C.byref(nb))
    return nb.value / sec.value / 1e9, sec.value * 1e6


print(f"config {cfg}: B-hat {p.resident_bytes()[1] / 1e9:.3f} GB; columns: U dbuf WGs/CU(cap) -> GB/s, us per pass")
for rep in range(2):
    for u, dbuf in ((4, 0), (4, 1), (8, 0), (8, 1), (16, 0), (16, 1), (32, 0), (8, 2), (16, 2), (16, 3)):   # dbuf & 2: XCD-contiguous runs
        row = []
        for wgs in (0, 8, 6, 5, 4, 3, 2, 1):
            g, us = run(u, dbuf, wgs)
            row.append(f"{wgs}:{g:.0f}")
        print(f"U={u:2d} dbuf={dbuf} " + " ".join(row), flush=True)
