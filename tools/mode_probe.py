#!/usr/bin/env python3
"""Is the ~177 / ~185 us bimodality of mac_rows at config 3 a property of the process (where the buffers landed) or of
the moment?  One process, 12 batches of 40 encrypts, kernel time per batch from the library's HIP events; optionally the
context (and with it every buffer) is rebuilt between batches.  Run on the GPU box: python tools/mode_probe.py [rebuild]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

import pvw_rs_amd as P  # noqa: E402
from pvw_rs_amd import _ffi, workloads as W  # noqa: E402

rebuild = len(sys.argv) > 1 and sys.argv[1] == "rebuild"
n, k, l, L, _ = W.ENCRYPT_CONFIGS["c3"]
dev = torch.device("cuda", 0)


def make():
    p = P.PvwParametersBuilder().set_parties(n).set_dimension(k).set_l(l).set_moduli(W.bench_moduli(L)).build()
    g = P.GlobalPublicKey.new(P.PvwCrs.new_deterministic(p, W.SEED_A))
    g.fill_uniform(W.SEED_B)
    return p, g


p, gpk = make()
scalars = torch.tensor(W.scalars(n), dtype=torch.int64, device=dev)
c1 = torch.zeros((k, L, l), dtype=torch.int64, device=dev)
c2 = torch.zeros((n, L, l), dtype=torch.int64, device=dev)
rnd = _ffi.pvw_randomness_t()
rnd.mode = _ffi.RND_SEED
C.memmove(rnd.seed, W.SEED_ENC, 32)
out = []
for batch in range(12):
    if rebuild and batch:
        del gpk, p
        p, gpk = make()
    def step():
        p._call("pvw_encrypt_device", C.c_void_p(scalars.data_ptr()), n, C.byref(rnd), C.c_void_p(c1.data_ptr()),
                C.c_void_p(c2.data_ptr()), P.REPR_NTT, None)
    for _ in range(10):
        step()
    p.synchronize()
    p.set_profiling(True)
    p.reset_profiling()
    for _ in range(40):
        step()
    p.synchronize()
    ms, cnt = p.kernel_time("mac_rows")
    p.set_profiling(False)
    out.append(round(ms / max(cnt, 1) * 1000, 1))
    time.sleep(0.05)
print(("rebuilt context per batch: " if rebuild else "one context: ") + " ".join(str(x) for x in out))
