#!/usr/bin/env python3
"""Is the placement regime of mac_rows (profiles/r03_mac_rows_regime_timeline.txt: an allocation is either ~5 % slow for
every XCD or not, for its whole life) a property any streaming kernel sees?  Pure torch: N buffers of 1.09 GiB held at
once, a read-only reduction timed on each with events, several rounds -- per-buffer medians and their spread.  If the
same buffers are slow in every round, a cheap read probe at allocation time could choose among candidates.
    python tools/alloc_regime.py [buffers=12] [rounds=4]"""
import sys

import torch

n_buf = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
words = (1090 << 20) // 8
bufs = [torch.ones(words, dtype=torch.int64, device=dev) for _ in range(n_buf)]
torch.cuda.synchronize()


def time_sum(x, reps=20):
    for _ in range(3):
        x.sum()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        x.sum()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3     # us


res = [[0.0] * rounds for _ in range(n_buf)]
for r in range(rounds):
    for i, x in enumerate(bufs):
        res[i][r] = time_sum(x)
print(f"{n_buf} buffers of {words * 8 / 2**30:.2f} GiB, torch.sum (read-only), us per pass; address; one row per buffer, one column per round")
for i, x in enumerate(bufs):
    print(f"buf {i:2d} @ {x.data_ptr():#014x}: " + " ".join(f"{v:7.1f}" for v in res[i]) + f"   GB/s {words * 8 / min(res[i]) / 1e3:7.1f}")
best = [min(r) for r in res]
print(f"fastest buffer {min(best):.1f} us, slowest {max(best):.1f} us, spread {(max(best) / min(best) - 1) * 100:.1f} %")
