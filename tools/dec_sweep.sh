#!/bin/bash
# the schedule selectors exist in the measurement build only (include/pvw_hip_tuning.h)
# sweep of decrypt_mac launch shapes: PVW_DEC_VARIANT x PVW_DEC_C on two decrypt workloads
out=gpurun_out/dec_sweep.log; : > $out
run() {
  echo "cfg $1 variant $2 c $3" >> $out
  PVW_DEC_VARIANT=$2 PVW_DEC_C=$3 timeout -k 10 200 python bench.py --tuning-library --path decrypt --config $1 --steps 30 --warmup 5 --no-worst-case 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  ms_per_step',round(d['ms_per_step'],4),'mac_us',round(d['roofline']['avg_launch_us'],1),'GB/s',round(d['roofline']['achieved']))
" >> $out || exit 1
}
# 60 = full-width form where the shape allows (config 5), 10 = dealer-grouped form; c = replicas over j (0 = by shape)
for v in 60 10; do for c in 0 1 2 3; do run c5shard $v $c; done; done
for c in 0 3 4 5 7; do run d3 10 $c; done
