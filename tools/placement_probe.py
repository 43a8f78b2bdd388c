#!/usr/bin/env python3
"""mac_rows (packed stream) at config 3 against WHERE the packed B-hat was allocated (tuning build): one context, the copy
re-packed after every refill of the public key, kernel time from the library's HIP events.
    python tools/placement_probe.py c3"""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

import pvw_rs_amd as P  # noqa: E402
from pvw_rs_amd import _ffi, workloads as W  # noqa: E402

_ffi.select("tuning")   # the measurement build: schedule selectors, stamps, probes

n, k, l, L, _ = W.ENCRYPT_CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
dev = torch.device("cuda", 0)
p = P.PvwParametersBuilder().set_parties(n).set_dimension(k).set_l(l).set_moduli(W.bench_moduli(L)).build()
gpk = P.GlobalPublicKey.new(P.PvwCrs.new_deterministic(p, W.SEED_A))
scalars = torch.tensor(W.scalars(n), dtype=torch.int64, device=dev)
c1 = torch.zeros((k, L, l), dtype=torch.int64, device=dev)
c2 = torch.zeros((n, L, l), dtype=torch.int64, device=dev)
rnd = _ffi.pvw_randomness_t()
rnd.mode = _ffi.RND_SEED
C.memmove(rnd.seed, W.SEED_ENC, 32)


def step():
    p._call("pvw_encrypt_device", C.c_void_p(scalars.data_ptr()), n, C.byref(rnd), C.c_void_p(c1.data_ptr()),
            C.c_void_p(c2.data_ptr()), P.REPR_NTT, None)


hold = []
for rnd_i in range(14):
    hold.append(torch.empty((64 << 20) * (1 + rnd_i % 3), dtype=torch.uint8, device=dev))   # shifts what the allocator hands out next
    gpk.fill_uniform(W.SEED_B)               # invalidates the packed copy: the next encrypt re-packs at the new offset
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
    print(f"re-pack {rnd_i:2d}: mac_rows {ms / max(cnt, 1) * 1000:6.1f} us")
