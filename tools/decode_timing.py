#!/usr/bin/env python3
"""Cycle counts of the phases of decode_chain_kernel (PVW_DECODE_TIMING, tuning aid; run on the GPU box):
    python tools/decode_timing.py L l count [random|dealt] [PVW_DECODE_VARIANT ...]
random = uniform residues (the longest path); dealt = message * Delta^j plus noise of the size a decrypt leaves."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
from pvw_rs_amd import workloads as M
import pvw_rs_amd as P
from pvw_rs_amd import _ffi

_ffi.select("tuning")      # PVW_DECODE_TIMING exists in the measurement build only

L, l, count = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
moduli = M.bench_moduli(L)
p = (P.PvwParametersBuilder().set_parties(3).set_dimension(4).set_l(l).set_moduli(moduli)
     .set_secret_variance(0.5).set_error_bounds(100, 200).build())
rng = np.random.default_rng(1)
kind = sys.argv[4] if len(sys.argv) > 4 else "random"
if kind == "dealt":
    Q = 1
    for q in moduli:
        Q *= q
    delta = p.delta()
    arr = np.zeros((count, L, l), dtype=np.uint64)
    for d in range(count):
        msg = int(rng.integers(0, 2 ** 62))
        z = [(-(msg * delta ** j) + int(rng.integers(-40000, 40001))) % Q for j in range(l)]
        for i, q in enumerate(moduli):
            arr[d, i] = [c % q for c in z]
else:
    arr = np.stack([rng.integers(0, q, size=(count, l), dtype=np.uint64) for q in moduli], axis=1)
print(f"input: {kind}")
for variant in sys.argv[5:] or ["0"]:
    os.environ["PVW_DECODE_VARIANT"] = variant
    for mode, name in ((1, "staging"), (2, "phase1 lifts + first division"), (3, "chain"), (4, "step: sub_centre"), (5, "step: 2|p|+Delta"), (6, "step: division"), (7, "phase 1: the lifts, between the barriers"), (8, "the first lift of wave 0"), (9, "short-cut candidates of wave 0"), (10, "Horner value + its full lift")):
        os.environ["PVW_DECODE_TIMING"] = str(mode)
        P.decode_scalar_pvw(p, arr)
        got = np.array(P.decode_scalar_pvw(p, arr), dtype=np.float64)
        print(f"variant {variant} {name}: median {np.median(got):.0f} min {got.min():.0f} max {got.max():.0f} ticks of clock64()")
    os.environ.pop("PVW_DECODE_TIMING")
