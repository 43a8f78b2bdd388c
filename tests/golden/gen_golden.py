"""Generates the golden fixtures under tests/golden/ from oracle/pvw_model.py.

The reference (gnosisguild/pvw-rs) holds no golden vectors and cannot be built or run in
this pipeline (Rust crate, un-vendored fhe-math), so these vectors come from this repo's
independent big-integer model (plain negacyclic schoolbook arithmetic in Z_Q[X]/(X^l+1)),
which is pinned against the properties the reference's own tests state
(tests/test_oracle_model.py).  Everything is stored as RNS residues in the power basis,
API layout [..][L][l] uint64, plus the small signed inputs.

    python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
sys.path.insert(0, os.path.join(HERE, ".."))

import pvw_model as M  # noqa: E402
from _util import EXAMPLE_MODULI, SEED, TEST_MODULI, make_system, ring_to_rns  # noqa: E402

CASES = {
    # name: (n, k, l, moduli, variance, bounds)
    "n3_k4_l8_test3": (3, 4, 8, TEST_MODULI, 0.5, None),            # tests/crypto.rs:56-70
    "n10_k4_l16_test3": (10, 4, 16, TEST_MODULI, 0.5, None),        # tests/crypto.rs:237-305
    "n30_k16_l32_test3": (30, 16, 32, TEST_MODULI, 0.5, None),      # tests/params.rs:37-53 at reduced k
    "n5_k16_l8_example4": (5, 16, 8, EXAMPLE_MODULI, 10.0, (1, 1172385)),  # examples/pvw_valid_dec.rs:40-52 at reduced k
    "n8_k16_l8_bench17": (8, 16, 8, M.bench_moduli(17), 0.5, (100, 200)),  # SURVEY 8(d) chain, 1037-bit Q
}


def main():
    for name, (n, k, l, moduli, var, bounds) in CASES.items():
        scalars = None
        if name.startswith("n8_"):
            scalars = [1, (1 << 64) - 5, 1 << 63, (1 << 63) - 1, 0, 123456789, 2 ** 32 - 1, 4242]
        s = make_system(n, k, l, moduli, variance=var, bounds=bounds, seed=SEED, scalars=scalars)
        P = s["P"]
        noisy = [M.decrypt_noisy(P, s["c1"], s["c2"][i], s["sk"][i]) for i in range(n)]
        decoded = [M.decode_scalar_pvw(z, P) for z in noisy]
        words = lambda x: np.array([(x >> (64 * i)) & ((1 << 64) - 1) for i in range((x.bit_length() + 63) // 64)], dtype=np.uint64)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            n=n, k=k, l=l, moduli=np.array(moduli, dtype=np.uint64), variance=var,
            bound1=P.error_bound_1, bound2=P.error_bound_2,
            delta_words=words(P.delta), q_words=words(P.Q),
            psi=np.array([M.minimal_primitive_root(q, 2 * l) for q in moduli], dtype=np.uint64),
            A_pb=s["A_pb"], B_pb=s["B_pb"],
            sk=np.array(s["sk"], dtype=np.int64), ek=np.array(s["ek"], dtype=np.int64),
            r=np.array(s["r"], dtype=np.int64), e1=np.array(s["e1"], dtype=np.int64), e2=np.array(s["e2"], dtype=np.int64),
            scalars=np.array(s["scalars"], dtype=np.uint64),
            c1_pb=s["c1_pb"], c2_pb=s["c2_pb"], g_pb=s["g_pb"],
            noisy_pb=np.array([ring_to_rns(z, moduli) for z in noisy], dtype=np.uint64),
            decoded=np.array(decoded, dtype=np.uint64),
        )
        print(name, "decoded ok:", decoded == [x if x < (1 << 63) else 0 for x in s["scalars"]] or decoded)


if __name__ == "__main__":
    main()
