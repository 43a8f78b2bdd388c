#!/bin/bash
# The judged evidence for the headline workload, all on ONE box in one gpurun: the default bench.py line, the rocprofv3 kernel trace and the
# two HBM-traffic passes of the same command (-> profiles/<tag>_c3_*), and the SQ counters + held clock of the mac_rows launch.
# usage (on the GPU box): bash tools/evidence.sh r03
set -e
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
cd $R && mkdir -p gpurun_out
timeout -k 10 400 python3 bench.py > gpurun_out/${tag}_c3_default_bench.json 2> gpurun_out/${tag}_c3_default_bench.err
bash tools/profile_all.sh ${tag}_c3 c3 > gpurun_out/${tag}_c3_profile.log 2>&1
bash tools/pmc_mac.sh > gpurun_out/${tag}_mac_rows_pmc.log 2>&1
cp gpurun_out/pmc_mac/summary.txt gpurun_out/${tag}_mac_rows_pmc.txt
cp profiles/${tag}_c3_summary.json profiles/${tag}_c3_kernel_stats.csv gpurun_out/ 2>/dev/null || true
python3 - <<PY
import json
d = json.loads(open("gpurun_out/${tag}_c3_default_bench.json").read().strip().splitlines()[-1])
s = json.load(open("profiles/${tag}_c3_summary.json"))
k = [v for n, v in s["kernels"].items() if n.startswith("mac_rows")][0]
print("bench line: value %.0f parties/s, %.1f us/step, mac_rows %.1f us (HIP events), frac %.3f, frac_of_streamed_bytes %.3f" % (
    d["value"], d["ms_per_step"] * 1e3, d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"]["frac_of_streamed_bytes"]))
print("rocprofv3:  mac_rows %.1f us avg over %d launches; HBM traffic %.4f GB per launch" % (k["avg_us"], k["calls"], k.get("hbm_traffic_bytes_per_launch", 0) / 1e9))
PY
