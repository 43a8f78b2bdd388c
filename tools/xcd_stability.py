#!/usr/bin/env python3
"""Is the per-XCD streaming rate of mac_rows a stable property of the box?  Tuning build, run on the GPU box:
    python tools/xcd_stability.py [config]
Prints, for 8 stamped launches (variant 40), the time each XCD finishes its share (the hardware deals workgroups
round-robin, 1/8 each) relative to the launch's mean."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import torch  # noqa: E402,F401

import pvw_rs_amd as P  # noqa: E402
from pvw_rs_amd import _ffi, workloads as W  # noqa: E402

_ffi.select("tuning")   # the measurement build: schedule selectors, stamps, probes

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
n, k, l, L, _ = W.ENCRYPT_CONFIGS[cfg]
p = (P.PvwParametersBuilder().set_parties(n).set_dimension(k).set_l(l).set_moduli(W.bench_moduli(L)).build())
gpk = P.GlobalPublicKey.new(P.PvwCrs.new_deterministic(p, W.SEED_A))
gpk.fill_uniform(W.SEED_B)
dev = torch.device("cuda", 0)
scalars = torch.tensor(W.scalars(n), dtype=torch.int64, device=dev)
c1 = torch.zeros((k, L, l), dtype=torch.int64, device=dev)
c2 = torch.zeros((n, L, l), dtype=torch.int64, device=dev)
rnd = _ffi.pvw_randomness_t()
rnd.mode = _ffi.RND_SEED
C.memmove(rnd.seed, W.SEED_ENC, 32)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def step():
    p._call("pvw_encrypt_device", C.c_void_p(scalars.data_ptr()), n, C.byref(rnd), C.c_void_p(c1.data_ptr()),
            C.c_void_p(c2.data_ptr()), P.REPR_NTT, stream)


os.environ["PVW_MAC_VARIANT"] = "40"
for _ in range(200):
    step()
torch.cuda.synchronize()
R = 128 // l
blocks = ((n + R - 1) // R + (k + R - 1) // R) * L
for rep in range(8):
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    st = np.zeros((blocks, 2), dtype=np.uint64)
    hw = np.zeros(blocks, dtype=np.uint32)
    p._call("pvw_tuning_read_stamps", st.ctypes.data_as(C.c_void_p), hw.ctypes.data_as(C.c_void_p), blocks)
    st = st.astype(np.int64)
    t0 = st[:, 0].min()
    end = (st[:, 1] - t0) / 100.0
    xcd = hw >> 28
    ends = [end[xcd == x].max() for x in range(8)]
    cnt = [int((xcd == x).sum()) for x in range(8)]
    by_block = [int(np.bincount(xcd[np.arange(blocks) % 8 == x], minlength=8).argmax()) for x in range(8)]
    print(f"launch {rep}: span {end.max():6.1f} us; XCD ends - mean: " + " ".join(f"{e - np.mean(ends):+5.1f}" for e in ends) +
          f"; items {cnt}; XCD of blocks b%8==x: {by_block}")
