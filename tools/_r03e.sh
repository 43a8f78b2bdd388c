mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "mac_rows or packed or prepare or switches or seed_mode or config3 or ragged or rccl or forced or golden or device_pointer or concurrent or sharded" > gpurun_out/r03e_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03e_pytest.log; tail -6 gpurun_out/r03e_pytest.log
J='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]), round(d["ms_per_step"]*1000,1), round(r["avg_launch_us"],1), r.get("kernel"), {k: round(v*1000,1) for k,v in d["kernel_ms_per_step"].items() if v})'
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --no-cpu --no-probe --sustain-seconds 0 2>/dev/null | python -c "$J" front1_shipped
  PVW_MAC_FRONT=0 timeout -k 10 200 python bench.py --tuning-library --no-cpu --no-probe --sustain-seconds 0 2>/dev/null | python -c "$J" front0_classic
  PVW_MAC_XBAL=16 timeout -k 10 200 python bench.py --tuning-library --no-cpu --no-probe --sustain-seconds 0 2>/dev/null | python -c "$J" front1_xbal16
  PVW_MAC_XBAL=10 timeout -k 10 200 python bench.py --tuning-library --no-cpu --no-probe --sustain-seconds 0 2>/dev/null | python -c "$J" front1_xbal10
  PVW_MAC_XBAL=24 PVW_MAC_FRONT=0 timeout -k 10 200 python bench.py --tuning-library --no-cpu --no-probe --sustain-seconds 0 2>/dev/null | python -c "$J" front0_xbal24
done 2>&1 | tee gpurun_out/r03e_front_ab.txt
for c in c2 c4shard ref128x; do
  timeout -k 10 300 python bench.py --no-cpu --no-probe --sustain-seconds 0 --config $c 2>/dev/null | python -c "$J" $c
  PVW_MAC_FRONT=0 timeout -k 10 300 python bench.py --tuning-library --no-cpu --no-probe --sustain-seconds 0 --config $c 2>/dev/null | python -c "$J" ${c}_classic
  PVW_MAC_XBAL=16 timeout -k 10 300 python bench.py --tuning-library --no-cpu --no-probe --sustain-seconds 0 --config $c 2>/dev/null | python -c "$J" ${c}_xbal16
done 2>&1 | tee gpurun_out/r03e_configs.txt
timeout -k 10 200 python tools/alloc_regime.py 14 3 > gpurun_out/r03e_alloc_regime.txt 2>&1; tail -20 gpurun_out/r03e_alloc_regime.txt
PVW_MAC_XBAL=16 timeout -k 10 300 python tools/regime_timeline.py 4 > gpurun_out/r03e_regime_xbal16.txt 2>&1; tail -12 gpurun_out/r03e_regime_xbal16.txt
