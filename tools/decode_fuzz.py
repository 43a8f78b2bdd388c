#!/usr/bin/env python3
"""Differential run of the on-device gadget decode against the host big-integer decode (pvw_decode_host) on inputs around
the short cuts of decode_chain_kernel (run on the GPU box):
    python tools/decode_fuzz.py [cases per class=3000] [seed=1]
Classes, at the bench chains (17 limbs l=8, 34 limbs l=16) and the reference's 4 x 56-bit chain: ciphertext-shaped inputs
z_j = -m Delta^j + n_j with small noise; one noise value anywhere up to Q; one residue of one coefficient replaced (the
value is then no longer small on that limb only); noise on the rounding boundaries of Delta; messages beyond 64 bits;
uniform residues.  Every mismatch is printed; exit status 1 if there is one."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import pvw_rs_amd as P  # noqa: E402
from pvw_rs_amd import workloads as W  # noqa: E402

per = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for l, moduli in ((8, W.bench_moduli(17)), (16, W.bench_moduli(34)), (8, W.REFERENCE_128_MODULI)):
    L = len(moduli)
    p = (P.PvwParametersBuilder().set_parties(3).set_dimension(4).set_l(l).set_moduli(moduli).build())
    Q, D = p.q_total(), p.delta()
    classes = ("small noise", "one wild noise", "one residue replaced", "rounding boundary", "wide message", "uniform")
    for cls in classes:
        zs = np.zeros((per, L, l), dtype=np.uint64)
        for c in range(per):
            m = int(rng.integers(0, 2 ** 63)) if cls != "wide message" else int(rng.integers(2 ** 63, 2 ** 64 - 1, dtype=np.uint64)) * int(rng.integers(1, 2 ** 20))
            noise = [int(x) for x in rng.integers(-60000, 60001, size=l)]
            if cls == "one wild noise":
                noise[int(rng.integers(0, l))] = int.from_bytes(rng.bytes(Q.bit_length() // 8 + 8), "little") % Q
            if cls == "rounding boundary":
                j = int(rng.integers(0, l))
                noise[j] = int(rng.choice([1, -1])) * (D // 2 + int(rng.integers(-2, 3)))
            z = [(-(m * D ** j) + noise[j]) % Q for j in range(l)]
            if cls == "uniform":
                z = [int.from_bytes(rng.bytes(Q.bit_length() // 8 + 8), "little") % Q for _ in range(l)]
            for i, q in enumerate(moduli):
                zs[c, i] = [v % q for v in z]
            if cls == "one residue replaced":
                i, j = int(rng.integers(0, L)), int(rng.integers(0, l))
                zs[c, i, j] = int(rng.integers(0, moduli[i]))
        dev = [int(x) for x in P.decode_scalar_pvw(p, zs)]
        host = [int(x) for x in P.decode_scalar_pvw_host(p, zs)]
        diff = [c for c in range(per) if dev[c] != host[c]]
        bad += len(diff)
        print(f"l={l} limbs={L} {cls:22s}: {per} inputs, {len(diff)} mismatches" + (f" first at {diff[0]}: device {dev[diff[0]]} host {host[diff[0]]}" if diff else ""), flush=True)
sys.exit(1 if bad else 0)
