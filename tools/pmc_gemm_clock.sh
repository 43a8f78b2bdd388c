#!/bin/bash
# Shader clock under the digit GEMM: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) and the MFMA-busy cycles next to
# the kernel's duration from the same run's kernel trace.  Run on the GPU box.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_gemm_clk && mkdir -p $R/gpurun_out/pmc_gemm_clk
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_gemm_clk/p -- python3 $R/bench.py --dealers ${1:-64} --steps 3 --warmup 1 --no-cpu --sustain-seconds 0 --no-probe > $R/gpurun_out/pmc_gemm_clk/p.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_gemm_clk/p/**/*_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0][:60]
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "Start_Timestamp" in r and r.get("End_Timestamp"):
        acc[n]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
kt = glob.glob("gpurun_out/pmc_gemm_clk/p/**/*_kernel_trace.csv", recursive=True)
dur = collections.defaultdict(list)
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Kernel_Name"].split("(")[0][:60]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for n, c in acc.items():
    if "gemm" not in n: continue
    d = dur.get(n) or c.get("_dur_ns") or [0]
    us = sum(d) / len(d) / 1e3
    g = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / 8
    m = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    print(f"{n}: {us:.1f} us, {g:.4g} cycles per XCD -> {g / us / 1e3:.2f} GHz, MFMA busy {m:.4g} / (1024 x cycles) = {m / 1024 / g:.3f}")
PY
