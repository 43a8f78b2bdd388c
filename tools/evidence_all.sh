#!/bin/bash
# Every bench line the round's documents quote, on ONE box in one gpurun: tools/evidence.sh (the headline workload with its
# rocprofv3 trace and counter passes) and one bench.py line per secondary configuration -> gpurun_out/<tag>_*_bench.json.
# usage (on the GPU box): bash tools/evidence_all.sh r03
set -e
tag=${1:-r03}
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
bash tools/evidence.sh $tag
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/${tag}_${name}_bench.json 2> gpurun_out/${tag}_${name}_bench.err; echo "$name done"; }
run c2 --config c2 --steps 200 --warmup 20 --no-cpu
run c4shard --config c4shard --steps 50 --warmup 5 --no-cpu
run c4full --config c4full --steps 10 --warmup 2 --no-cpu
run ref128x --config ref128x --steps 100 --warmup 10 --no-cpu
PVW_MAC_PACKED=0 run ref128x_unpacked --config ref128x --steps 100 --warmup 10 --no-cpu --tuning-library
run multi64 --dealers 64 --steps 20 --warmup 3 --no-cpu
run ref128x_multi64 --config ref128x --dealers 64 --steps 20 --warmup 3 --no-cpu
run keygen --path keygen --steps 20 --warmup 3 --no-cpu
bash tools/profile_all.sh ${tag}_c5shard c5shard --path decrypt --config c5shard --no-worst-case > gpurun_out/${tag}_c5shard_profile.log 2>&1 || echo "c5shard profile failed"
cp profiles/${tag}_c5shard_summary.json profiles/${tag}_c5shard_kernel_stats.csv gpurun_out/ 2>/dev/null || true
run decrypt_c5shard --path decrypt --config c5shard --steps 200 --warmup 20
run decrypt_c5one --path decrypt --config c5one --steps 300 --warmup 30
run decrypt_c5full --path decrypt --config c5full --steps 20 --warmup 3
run decrypt_d3 --path decrypt --config d3 --steps 200 --warmup 20
python3 - <<PY
import glob, json
for f in sorted(glob.glob("gpurun_out/${tag}_*_bench.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "UNREADABLE", e)
        continue
    w = j.get("worst_case_random_residues")
    print("%-46s %12.0f %s  %8.1f us/step  frac %.3f%s" % (f.split("/")[-1], j["value"], j["unit"], j["ms_per_step"] * 1e3, j["roofline"]["frac"],
          ("  | uniform residues %.1f us/step" % (w["ms_per_step"] * 1e3)) if w else ""))
PY
