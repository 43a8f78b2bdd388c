#!/bin/bash
# rocprofv3 kernel trace of one bench command -> profiles/<tag>_kernel_stats.csv + <tag>_summary.json (no PMC passes)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_$tag && mkdir -p $R/gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag/trace -- python3 $R/bench.py "$@" > $R/gpurun_out/prof_$tag/trace.log 2>&1
cd $R && python3 tools/summarize_prof.py $tag gpurun_out/prof_$tag/trace > /dev/null
