#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (kernel-trace --stats, and separate --pmc passes)
into the small text/JSON summaries committed under profiles/.

    python tools/summarize_prof.py <round-tag> <trace_dir> [<pmc_fetch_dir> <pmc_write_dir>] [--config c3]

HBM traffic per launch follows MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and
WRITE_SIZE are collected in separate passes, are in KiB, and on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide (16 B/lane) coalesced stream, so it is doubled.
"""
import collections
import csv
import glob
import json
import os
import sys


def stats(trace_dir):
    f = glob.glob(os.path.join(trace_dir, "**", "*_kernel_stats.csv"), recursive=True)[0]
    return list(csv.DictReader(open(f)))


def pmc(dirname, counter):
    f = glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    config = "c3"
    if "--config" in sys.argv:
        config = sys.argv[sys.argv.index("--config") + 1]
    tag, trace_dir = args[0], args[1]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    rows = stats(trace_dir)
    with open(os.path.join(root, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
    out = {"tag": tag, "config": config, "kernels": {}}
    for r in rows:
        if "pvw::" in r["Name"]:
            short = r["Name"].split("pvw::")[1].split("(")[0]
            out["kernels"][short] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                     "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                                     "pct": float(r["Percentage"])}
    if len(args) >= 4:
        fetch, write = pmc(args[2], "FETCH_SIZE"), pmc(args[3], "WRITE_SIZE")
        for name, (fv, cnt) in fetch.items():
            if "pvw::" not in name:
                continue
            short = name.split("pvw::")[1].split("(")[0]
            wv = write.get(name, (0.0, 0))[0]
            k = out["kernels"].setdefault(short, {})
            k.update({"FETCH_SIZE_KiB_raw": fv, "WRITE_SIZE_KiB": wv, "pmc_launches": cnt,
                      "hbm_traffic_bytes_per_launch": (2.0 * fv + wv) * 1024.0,
                      "note": "traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024; FETCH_SIZE doubled per the gfx950 correction"})
    # the bench line the traced process itself printed (its HIP-event timing of the very launches the trace holds)
    log = os.path.join(os.path.dirname(os.path.normpath(trace_dir)), "trace.log")
    if os.path.exists(log):
        for line in open(log, errors="replace"):
            if line.startswith("{") and '"metric"' in line:
                try:
                    j = json.loads(line)
                    out["traced_process_bench_line"] = {
                        "ms_per_step": j.get("ms_per_step"), "kernel_ms_per_step": j.get("kernel_ms_per_step"),
                        "dominant_kernel_avg_launch_us_by_hip_events": j.get("roofline", {}).get("avg_launch_us"),
                        "note": "same process as the kernel trace above (event timing runs under the tracer here)"}
                except ValueError:
                    pass
    with open(os.path.join(root, f"{tag}_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
