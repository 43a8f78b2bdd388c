#!/bin/bash
# A/B the streaming-schedule variants of mac_rows / decrypt_mac on the GPU box (tuning aid).
# usage: tools/sweep_variants.sh <outfile>
out=${1:-gpurun_out/sweep.txt}
: > $out
for c in c3 c2 c4shard; do
  for v in 0 3 7 8; do
    line=$(PVW_MAC_VARIANT=$v timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu --config $c 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step']*1000,1), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['achieved']))")
    echo "mac $c variant=$v parties/s,us/step,mac_us,GB/s: $line" | tee -a $out
  done
done
for c in; do
  for v in 10; do
    line=$(PVW_DEC_VARIANT=$v timeout -k 10 120 python bench.py --path decrypt --steps 20 --warmup 3 --config $c 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step']*1000,1), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['achieved']))")
    echo "dec $c variant=$v ct/s,us/step,mac_us,GB/s: $line" | tee -a $out
  done
done
