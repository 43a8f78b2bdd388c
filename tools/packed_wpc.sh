#!/bin/bash
# packed mac_rows: register allocation for 2 / 3 / 4 workgroups per CU (tuning build rebuilt on the GPU box per setting)
out=${1:-gpurun_out/packed_wpc.txt}; : > $out
export PVW_HIP_LIBRARY=tuning
for wpc in 2 3 4; do
  PVW_PACKED_WPC=$wpc python pvw_rs_amd/build.py --force --tuning-only --quiet > /dev/null 2>&1 || { echo "build failed for $wpc" >> $out; continue; }
  for c in c3 c3 c4shard c2; do
    line=$(timeout -k 10 200 python bench.py --config $c --no-cpu --no-probe --sustain-seconds 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step']*1000,2), round(d['roofline']['avg_launch_us'],2), round(d['roofline']['frac'],4))")
    echo "wpc=$wpc $c: $line" | tee -a $out
  done
done
python pvw_rs_amd/build.py --force --tuning-only --quiet > /dev/null 2>&1
