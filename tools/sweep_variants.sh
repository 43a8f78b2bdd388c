#!/bin/bash
# the schedule selectors exist in the measurement build only (include/pvw_hip_tuning.h)
# A/B the streaming-schedule variants of mac_rows on the GPU box (tuning aid).
# usage: [MACV="0 17"] [PACKED="1 0"] [CFGS="c3 c2 c4shard"] [REPS=2] tools/sweep_variants.sh <outfile>
# PVW_MAC_PACKED=1: the packed stream (variant 0 only: an explicit schedule of the unpacked kernel switches it off);
# PVW_MAC_VARIANT 0 / 17: the interleaved / the non-interleaved schedule of the unpacked kernel
out=${1:-gpurun_out/sweep.txt}
: > $out
for rep in $(seq 1 ${REPS:-1}); do
for c in ${CFGS:-c3 c2 c4shard}; do
  for pk in ${PACKED:-1 0}; do
  for v in ${MACV:-0 17}; do
    [ "$pk" = 1 ] && [ "$v" != 0 ] && continue
    line=$(PVW_MAC_PACKED=$pk PVW_MAC_VARIANT=$v timeout -k 10 120 python bench.py --tuning-library --steps ${STEPS:-40} --warmup 5 --no-cpu --no-probe --sustain-seconds 0 --config $c 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step']*1000,1), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['achieved']))")
    echo "mac $c packed=$pk variant=$v parties/s,us/step,mac_us,GB/s: $line" | tee -a $out
  done
  done
done
done
