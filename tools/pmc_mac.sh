#!/bin/bash
# SQ counters of the mac_rows launch that bench.py's default workload runs (the packed-stream kernel) + the shader clock it
# holds; run on the GPU box.  usage: tools/pmc_mac.sh [bench args...]   -> gpurun_out/pmc_mac/summary.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_mac && mkdir -p $R/gpurun_out/pmc_mac
B="python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --sustain-seconds 0 --no-probe"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_mac/p1 -- $B "$@" > $R/gpurun_out/pmc_mac/p1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_mac/p2 -- $B "$@" > $R/gpurun_out/pmc_mac/p2.log 2>&1
cd $R
python3 - <<'PY' | tee gpurun_out/pmc_mac/summary.txt
import csv, glob, collections
def is_mac(n):
    return "mac_rows" in n and "multi" not in n
acc = collections.defaultdict(list)
names = set()
for p in ("p1", "p2"):
    f = glob.glob(f"gpurun_out/pmc_mac/{p}/**/*_counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if is_mac(r["Kernel_Name"]):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            names.add(r["Kernel_Name"].split("(")[0])
dur = []
for f in glob.glob("gpurun_out/pmc_mac/p2/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if is_mac(r["Kernel_Name"]):
            dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print("# kernel:", ", ".join(sorted(names)))
for k in sorted(acc):
    v = acc[k]
    print(f"{k} avg per launch {sum(v) / len(v):.4g} launches {len(v)}")
if dur and acc.get("GRBM_GUI_ACTIVE"):
    us = sum(dur) / len(dur) / 1e3
    g = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"]) / 8
    print(f"# duration under the counter pass {us:.1f} us; GRBM_GUI_ACTIVE / 8 XCDs = {g:.4g} cycles -> {g / us / 1e3:.2f} GHz held")
if acc.get("SQ_ACTIVE_INST_VALU") and acc.get("SQ_BUSY_CYCLES") and acc.get("SQ_WAVE_CYCLES"):
    valu = sum(acc["SQ_ACTIVE_INST_VALU"]) / len(acc["SQ_ACTIVE_INST_VALU"])
    busy = sum(acc["SQ_BUSY_CYCLES"]) / len(acc["SQ_BUSY_CYCLES"])
    wavec = sum(acc["SQ_WAVE_CYCLES"]) / len(acc["SQ_WAVE_CYCLES"])
    # SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles; SQ_BUSY_CYCLES is summed over the 32 shader engines
    cyc = busy / 32
    print(f"# cycles per launch (SQ_BUSY_CYCLES / 32 SEs) {cyc:.4g}; VALU busy per SIMD = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles) = {4 * valu / 1024 / cyc:.3f}; "
          f"waves resident per SIMD = 4 x SQ_WAVE_CYCLES / (1024 x cycles) = {4 * wavec / 1024 / cyc:.2f}")
PY
