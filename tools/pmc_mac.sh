#!/bin/bash
# SQ counters of mac_rows (one pass of 8 counters); run on the GPU box.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_mac && mkdir -p $R/gpurun_out/pmc_mac
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_mac/p1 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --sustain-seconds 0 --no-probe "$@" > $R/gpurun_out/pmc_mac/p1.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_mac/p1/**/*_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "mac_rows" in r["Kernel_Name"] and "multi" not in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, "avg per launch %.4g" % (sum(v) / len(v)), "launches", len(v))
PY
