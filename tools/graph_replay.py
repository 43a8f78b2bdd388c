#!/usr/bin/env python3
"""Eager launches against a captured hipGraph replayed (run on the GPU box):
    python tools/graph_replay.py
for one encrypt at config 3 (prologue + mac_rows) and one decrypt_party_value at the config-5 geometry (key NTT, inner
products in ranges, range sums + inverse transform, decode, wipe): microseconds per call, 300 calls back to back."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

import pvw_rs_amd as P  # noqa: E402
from pvw_rs_amd import _ffi, workloads as W  # noqa: E402

dev = torch.device("cuda", 0)
lib = _ffi.lib()


def ptr(t):
    return C.c_void_p(t.data_ptr())


def timed(fn, reps=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def both(name, call, stream):
    eager = timed(lambda: call(C.c_void_p(stream.cuda_stream)))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        call(C.c_void_p(torch.cuda.current_stream().cuda_stream))
    rep = timed(g.replay)
    print(f"{name}: eager {eager:.1f} us per call, graph replay {rep:.1f} us", flush=True)


# encrypt, config 3
n, k, l, L, _ = W.ENCRYPT_CONFIGS["c3"]
p = P.PvwParametersBuilder().set_parties(n).set_dimension(k).set_l(l).set_moduli(W.bench_moduli(L)).build()
gpk = P.GlobalPublicKey.new(P.PvwCrs.new_deterministic(p, W.SEED_A))
gpk.fill_uniform(W.SEED_B)
s = torch.cuda.Stream(device=dev)
p.prepare(P.PREPARE_PACKED, s.cuda_stream)
scal = torch.tensor(W.scalars(n), dtype=torch.int64, device=dev)
c1 = torch.zeros((k, L, l), dtype=torch.int64, device=dev)
c2 = torch.zeros((n, L, l), dtype=torch.int64, device=dev)
rnd = _ffi.pvw_randomness_t()
rnd.mode = _ffi.RND_SEED
C.memmove(rnd.seed, W.SEED_ENC, 32)
both("encrypt n=4096 k=256 l=8 17 limbs", lambda st: P.api._check(lib.pvw_encrypt_device(p._h, ptr(scal), n, C.byref(rnd), ptr(c1), ptr(c2), P.REPR_NTT, st)), s)
del gpk, p

# one decrypt_party_value, config-5 geometry
D, k, l, L, _ = W.DECRYPT_CONFIGS["c5one"]
p = P.PvwParametersBuilder().set_parties(4).set_dimension(k).set_l(l).set_moduli(W.bench_moduli(L)).build()
g0 = torch.Generator(device=dev)
g0.manual_seed(5)
qmin = int(min(W.bench_moduli(L)))
c1s = torch.randint(0, qmin, (D, k, L, l), dtype=torch.int64, device=dev, generator=g0)
c2c = torch.randint(0, qmin, (D, L, l), dtype=torch.int64, device=dev, generator=g0)
sk = torch.from_numpy(p.sample_vec_cbd(W.SEED_ENC, P.DOM_SK, 0, k)).to(dev)
noisy = torch.zeros((D, L, l), dtype=torch.int64, device=dev)
vals = torch.zeros(D, dtype=torch.int64, device=dev)
both("decrypt_party_value k=512 l=16 34 limbs (uniform residues)",
     lambda st: P.api._check(lib.pvw_decrypt_batch_device(p._h, ptr(sk), ptr(c1s), ptr(c2c), D, P.REPR_NTT, ptr(noisy), ptr(vals), st)), s)
