#!/usr/bin/env python3
"""Per-workgroup timeline of one mac_rows launch (tuning build; run on the GPU box):
    python tools/mac_timeline.py [config] [variant 40|44]      (40: the unpacked kernel, 44: the packed stream)
Every workgroup stamps its first and last instruction (100 MHz counter); this prints the launch span, the ramp
(first start -> all slots busy), the drain (queue empty -> last end), the workgroup-duration distribution and the
number of resident workgroups over time."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import torch  # noqa: E402,F401  (device memory only)

import pvw_rs_amd as P  # noqa: E402
from pvw_rs_amd import _ffi, workloads as W  # noqa: E402

_ffi.select("tuning")   # the measurement build: schedule selectors, stamps, probes

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
variant = sys.argv[2] if len(sys.argv) > 2 else "40"
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


os.environ["PVW_MAC_VARIANT"] = variant
for _ in range(200):                     # warm: clocks up
    step()
torch.cuda.synchronize()
R = 128 // l
blocks = ((n + R - 1) // R + (k + R - 1) // R) * L
runs = []
for rep in range(5):
    step()
    torch.cuda.synchronize()
    st = np.zeros((blocks, 2), dtype=np.uint64)
    hw = np.zeros(blocks, dtype=np.uint32)
    p._call("pvw_tuning_read_stamps", st.ctypes.data_as(C.c_void_p), hw.ctypes.data_as(C.c_void_p), blocks)
    runs.append((st.astype(np.int64), hw))
st, hw = runs[-1]
t0 = st[:, 0].min()
start, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0      # microseconds
dur = end - start
span = end.max()
print(f"config {cfg} variant {variant}: {blocks} workgroups, span first start -> last end {span:.1f} us "
      f"(spans of 5 launches: {[round(float((r[0][:, 1].max() - r[0][:, 0].min()) / 100.0), 1) for r in runs]})")
print(f"workgroup duration us: min {dur.min():.1f} p10 {np.percentile(dur, 10):.1f} median {np.median(dur):.1f} "
      f"p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f}")
order = np.argsort(start)
print(f"starts: first {start.min():.2f}, 256th {np.sort(start)[min(255, blocks - 1)]:.2f}, 768th {np.sort(start)[min(767, blocks - 1)]:.2f}, "
      f"1024th {np.sort(start)[min(1023, blocks - 1)]:.2f}, last {start.max():.1f} us")
ev = np.concatenate([np.stack([start, np.ones_like(start)], 1), np.stack([end, -np.ones_like(end)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
resident = np.cumsum(ev[:, 1])
peak = resident.max()
print(f"resident workgroups: peak {int(peak)}")
grid = np.arange(0.0, span + 2.5, 2.5)
idx = np.searchsorted(ev[:, 0], grid, side="right") - 1
line = [int(resident[i]) if i >= 0 else 0 for i in idx]
print("resident every 2.5 us:", line)
last_start = start.max()
print(f"queue empty (last start) at {last_start:.1f} us; drain = {span - last_start:.1f} us; "
      f"workgroups still running then: {(end > last_start).sum()}")
busy = np.trapezoid(np.interp(grid, ev[:, 0], resident), grid) / (span * peak)
print(f"mean residency over the span: {busy:.3f} of peak")
early, late = dur[order[: blocks // 8]], dur[order[-blocks // 8:]]
print(f"duration of the first eighth {early.mean():.1f} us, of the last eighth {late.mean():.1f} us")
xcc = (hw >> 28) & 0xF                                              # XCC_ID in the top nibble
for x in range(8):
    sel = xcc == x
    if sel.any():
        print(f"XCD {x}: {int(sel.sum())} items, first start {start[sel].min():.1f}, last start {start[sel].max():.1f}, "
              f"last end {end[sel].max():.1f} us, median duration {np.median(dur[sel]):.1f} us")
