#!/usr/bin/env python3
"""Placement regimes of mac_rows (packed stream) at config 3, looked at per XCD (tuning build; run on the GPU box):
    python tools/regime_timeline.py [contexts=10]
The kernel time of the packed mac_rows depends on which allocation holds the packed B-hat (profiles/r02_mac_rows_placement.txt:
~177-180 us or ~185-188 us, stable for the allocation's life).  This builds the context `contexts` times in one process (every
build allocates anew), times 40 launches from the library's HIP events, then stamps one launch per workgroup (schedule 44) and
prints, per XCD, the stream rate while its queue was full, the median workgroup duration and when it ran dry -- so that a slow
regime can be read as "one XCD pair / stack late" or "everything uniformly slower"."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

import pvw_rs_amd as P  # noqa: E402
from pvw_rs_amd import _ffi, workloads as W  # noqa: E402

_ffi.select("tuning")
contexts = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n, k, l, L, _ = W.ENCRYPT_CONFIGS["c3"]
dev = torch.device("cuda", 0)
scalars = torch.tensor(W.scalars(n), dtype=torch.int64, device=dev)
c1 = torch.zeros((k, L, l), dtype=torch.int64, device=dev)
c2 = torch.zeros((n, L, l), dtype=torch.int64, device=dev)
rnd = _ffi.pvw_randomness_t()
rnd.mode = _ffi.RND_SEED
C.memmove(rnd.seed, W.SEED_ENC, 32)
R = 128 // l
blocks = ((n + R - 1) // R + (k + R - 1) // R) * L
hold = []
rows = []
for ci in range(contexts):
    p = P.PvwParametersBuilder().set_parties(n).set_dimension(k).set_l(l).set_moduli(W.bench_moduli(L)).build()
    gpk = P.GlobalPublicKey.new(P.PvwCrs.new_deterministic(p, W.SEED_A))
    gpk.fill_uniform(W.SEED_B)

    def step():
        p._call("pvw_encrypt_device", C.c_void_p(scalars.data_ptr()), n, C.byref(rnd), C.c_void_p(c1.data_ptr()),
                C.c_void_p(c2.data_ptr()), P.REPR_NTT, None)

    os.environ["PVW_MAC_VARIANT"] = "0"
    step()                                      # builds the packed copies
    p.synchronize()
    p.set_profiling(True)
    p.reset_profiling()
    for _ in range(4):
        step()
    p.synchronize()
    ems, ecnt = p.kernel_time("mac_rows")
    p.set_profiling(False)
    early = ems / max(ecnt, 1) * 1000           # what a trial at build time would see: the first four launches
    for _ in range(60):
        step()
    p.synchronize()
    p.set_profiling(True)
    p.reset_profiling()
    for _ in range(40):
        step()
    p.synchronize()
    ms, cnt = p.kernel_time("mac_rows")
    p.set_profiling(False)
    us = ms / max(cnt, 1) * 1000
    os.environ["PVW_MAC_VARIANT"] = "44"
    per_xcd = []
    spans = []
    for rep in range(3):
        for _ in range(3):
            step()
        p.synchronize()
        st = np.zeros((blocks, 2), dtype=np.uint64)
        hw = np.zeros(blocks, dtype=np.uint32)
        p._call("pvw_tuning_read_stamps", st.ctypes.data_as(C.c_void_p), hw.ctypes.data_as(C.c_void_p), blocks)
        st = st.astype(np.int64)
        t0 = st[:, 0].min()
        start, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0
        xcc = (hw >> 28) & 0xF
        share = (hw >> 24) & 0x7                                    # block id modulo 8
        share_match = float((xcc == share).mean())
        spans.append(float(end.max()))
        row = []
        for x in range(8):
            sel = xcc == x
            if not sel.any():
                row.append((0, 0.0, 0.0, 0.0))
                continue
            # stream rate of the XCD while it still had work queued: bytes of the workgroups that ended before its last start
            ls = start[sel].max()
            done = sel & (end <= ls)
            rate = done.sum() * k * 1024 * 61 / 64 / max(ls, 1e-9) / 1e6          # TB/s
            row.append((int(sel.sum()), float(np.median((end - start)[sel])), float(end[sel].max()), float(rate)))
        per_xcd.append(row)
    os.environ["PVW_MAC_VARIANT"] = "0"
    row = per_xcd[-1]
    print(f"context {ci:2d}: mac_rows {us:6.1f} us (first four launches {early:6.1f}) | stamped spans {[round(s, 1) for s in spans]} | XCC_ID == block id % 8 for {share_match * 100:.1f} % of the workgroups")
    print("    XCD:            " + " ".join(f"{x:7d}" for x in range(8)))
    print("    median wg us:   " + " ".join(f"{r[1]:7.1f}" for r in row))
    print("    runs dry at us: " + " ".join(f"{r[2]:7.1f}" for r in row))
    print("    TB/s while fed: " + " ".join(f"{r[3]:7.3f}" for r in row) + f"   sum {sum(r[3] for r in row):.2f}")
    rows.append((us, row))
    # keep something allocated between contexts so that the next build does not simply get the same pages back
    hold.append(torch.empty((96 << 20) * (1 + ci % 3), dtype=torch.uint8, device=dev))
    del gpk, p
fast = [r for r in rows if r[0] < 182.0]
slow = [r for r in rows if r[0] >= 182.0]
for name, grp in (("fast", fast), ("slow", slow)):
    if grp:
        med = np.median(np.array([[x[1] for x in r[1]] for r in grp]), axis=0)
        dry = np.median(np.array([[x[2] for x in r[1]] for r in grp]), axis=0)
        rate = np.median(np.array([[x[3] for x in r[1]] for r in grp]), axis=0)
        print(f"{name} regime ({len(grp)} contexts, kernel {np.median([r[0] for r in grp]):.1f} us): median wg us per XCD {np.round(med, 1).tolist()}, "
              f"dry at {np.round(dry, 1).tolist()}, TB/s while fed {np.round(rate, 3).tolist()}")
