#!/bin/bash
# rocprofv3 kernel trace + the two PMC passes for one bench command; run on the GPU box.
# usage: tools/profile_all.sh <tag> <config-for-summary> <bench args...>
set -e
tag=$1; cfg=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_$tag
mkdir -p $R/gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag/trace -- python3 $R/bench.py --no-cpu --sustain-seconds 0 --no-probe --steps 20 --warmup 3 "$@" > $R/gpurun_out/prof_$tag/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$tag/fetch -- python3 $R/bench.py --no-cpu --sustain-seconds 0 --no-probe --steps 5 --warmup 1 "$@" > $R/gpurun_out/prof_$tag/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$tag/write -- python3 $R/bench.py --no-cpu --sustain-seconds 0 --no-probe --steps 5 --warmup 1 "$@" > $R/gpurun_out/prof_$tag/write.log 2>&1
cd $R && python3 tools/summarize_prof.py $tag gpurun_out/prof_$tag/trace gpurun_out/prof_$tag/fetch gpurun_out/prof_$tag/write --config $cfg
