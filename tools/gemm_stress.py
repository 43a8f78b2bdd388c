#!/usr/bin/env python3
"""Race screen for the digit GEMM (the ping-pong form keeps LDS-DMA stages in flight across raw barriers: a
misplaced wait would show as a rare wrong tile).  Runs the same 64-dealer encrypt and the same key generation many
times on the SHIPPED library and compares every result bit for bit with the first one; the first is what the parity
tests check against the oracle.  Run on the GPU box:
    python tools/gemm_stress.py [iterations] [config]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import torch  # noqa: E402

import pvw_rs_amd as P  # noqa: E402
from pvw_rs_amd import _ffi, workloads as W  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = sys.argv[2] if len(sys.argv) > 2 else "c3"
n, k, l, L, _ = W.ENCRYPT_CONFIGS[cfg]
Dm = 64
params = P.PvwParametersBuilder().set_parties(n).set_dimension(k).set_l(l).set_moduli(W.bench_moduli(L)).build()
h, lib = params._h, _ffi.lib()
gpk = P.GlobalPublicKey.new(P.PvwCrs.new_deterministic(params, W.SEED_A))
gpk.fill_uniform(W.SEED_B)
dev = torch.device("cuda", 0)
scalars = torch.tensor(W.scalars(n), dtype=torch.int64, device=dev).repeat(Dm, 1).contiguous()
c1 = torch.zeros((Dm, k, L, l), dtype=torch.int64, device=dev)
c2 = torch.zeros((Dm, n, L, l), dtype=torch.int64, device=dev)
seeds = np.concatenate([np.frombuffer(P.api._dealer_seed(W.SEED_ENC, d), dtype=np.uint8) for d in range(Dm)]).copy()
# an explicit stream shared by torch and the library: a NULL stream argument selects the context's own (non-blocking)
# stream, which is not ordered against torch's default stream
ts = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(ts)
stream = C.c_void_p(ts.cuda_stream)


def enc():
    rc = lib.pvw_encrypt_multi_device(h, C.c_void_p(scalars.data_ptr()), Dm, n, seeds.ctypes.data_as(C.c_void_p),
                                      C.c_void_p(c1.data_ptr()), C.c_void_p(c2.data_ptr()), P.REPR_NTT, stream)
    assert rc == 0, _ffi.last_error()
    torch.cuda.synchronize()


enc()
ref1, ref2 = c1.clone(), c2.clone()
bad = 0
for it in range(iters):
    c1.zero_()
    c2.zero_()
    enc()
    if not (torch.equal(c1, ref1) and torch.equal(c2, ref2)):
        bad += 1
        d = (c2 != ref2).nonzero()
        print(f"iteration {it}: {int((c1 != ref1).sum())} c1 words and {int((c2 != ref2).sum())} c2 words differ; first c2 index {d[0].tolist() if len(d) else None}")
print(f"multi-dealer encrypt x{Dm}, {cfg}: {iters} repeats, {bad} differ from the first")

# key generation: the tiled public key after every call
sk = np.zeros((n, k, l), dtype=np.int64)
P.api._check(lib.pvw_sample_secret_keys(h, np.frombuffer(W.SEED_ENC, dtype=np.uint8).ctypes.data_as(C.c_void_p), 0, n,
                                        sk.ctypes.data_as(C.c_void_p)))
seed = np.frombuffer(W.SEED_B, dtype=np.uint8).copy()
kit = max(10, iters // 10)
first = None
badk = 0
for it in range(kit):
    P.api._check(lib.pvw_keygen(h, 0, n, sk.ctypes.data_as(C.c_void_p), None, seed.ctypes.data_as(C.c_void_p)))
    rows = gpk.matrix(0, 64, P.REPR_NTT), gpk.matrix(n - 64, n, P.REPR_NTT), gpk.matrix(n // 2, n // 2 + 64, P.REPR_NTT)
    cur = np.concatenate([r.ravel() for r in rows])
    if first is None:
        first = cur
    elif not np.array_equal(cur, first):
        badk += 1
        print(f"keygen iteration {it}: {int((cur != first).sum())} words differ")
print(f"key generation, {n} parties: {kit} repeats (three 64-party windows compared), {badk} differ from the first")
sys.exit(1 if bad or badk else 0)
