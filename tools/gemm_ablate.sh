#!/bin/bash
# Timing ablations of gemm_digits (tuning build, rebuilt ON the GPU box per ablation; results of the ablated builds
# are wrong by construction): 0 = as shipped, 8 = no A (raw tile) loads, 16 = no B (digit tile) loads, 24 = neither,
# 32 = no MFMA.  usage: tools/gemm_ablate.sh <outfile> [dealers]
out=${1:-gpurun_out/gemm_ablate.txt}; D=${2:-64}
: > $out
export PVW_HIP_LIBRARY=tuning
for abl in ${ABLS:-0 8 16 24 32}; do
  PVW_GEMM_ABLATE=$abl python pvw_rs_amd/build.py --force --tuning-only --quiet > /dev/null 2>&1 || { echo "build failed for $abl" >> $out; continue; }
  for dbg in 0 ${DBGS:-}; do
    line=$(PVW_GEMM_DEBUG=$dbg timeout -k 10 200 python bench.py --dealers $D --steps 10 --warmup 2 --no-cpu --no-probe --sustain-seconds 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms_per_step'].items() if v})")
    echo "ablate=$abl debug=$dbg dealers=$D ms/step, kernels: $line" | tee -a $out
  done
done
python pvw_rs_amd/build.py --force --tuning-only --quiet > /dev/null 2>&1
