#!/bin/bash
# SQ counters of the digit-GEMM kernel (two passes of 8 counters); run on the GPU box.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_gemm && mkdir -p $R/gpurun_out/pmc_gemm
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pmc_gemm/p1 -- python3 $R/bench.py --dealers 64 --steps 3 --warmup 1 --no-cpu --sustain-seconds 0 --no-probe > $R/gpurun_out/pmc_gemm/p1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_gemm/p2 -- python3 $R/bench.py --dealers 64 --steps 3 --warmup 1 --no-cpu --sustain-seconds 0 --no-probe > $R/gpurun_out/pmc_gemm/p2.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    f = glob.glob(f"gpurun_out/pmc_gemm/{p}/**/*_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm_digits" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(p, k, "avg per launch %.4g" % (sum(v) / len(v)), "launches", len(v))
PY
