mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "mac_rows or packed or prepare or switches or seed_mode or pinned or rccl or forced" > gpurun_out/r03f_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03f_pytest.log; tail -6 gpurun_out/r03f_pytest.log
J='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]), round(d["ms_per_step"]*1000,1), round(r["avg_launch_us"],1), r.get("kernel"), {k: round(v*1000,1) for k,v in d["kernel_ms_per_step"].items() if v})'
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --no-cpu --no-probe --sustain-seconds 0 2>/dev/null | python -c "$J" front1_shipped
  PVW_MAC_FRONT=0 timeout -k 10 200 python bench.py --tuning-library --no-cpu --no-probe --sustain-seconds 0 2>/dev/null | python -c "$J" front0_classic
  PVW_MAC_FRONT=2 timeout -k 10 200 python bench.py --tuning-library --no-cpu --no-probe --sustain-seconds 0 2>/dev/null | python -c "$J" front2_fused
done 2>&1 | tee gpurun_out/r03f_front_ab.txt
for c in c2 c4shard ref128x; do
  timeout -k 10 300 python bench.py --no-cpu --no-probe --sustain-seconds 0 --config $c 2>/dev/null | python -c "$J" $c
  PVW_MAC_FRONT=0 timeout -k 10 300 python bench.py --tuning-library --no-cpu --no-probe --sustain-seconds 0 --config $c 2>/dev/null | python -c "$J" ${c}_classic
done 2>&1 | tee gpurun_out/r03f_configs.txt
timeout -k 10 300 python bench.py --no-probe --sustain-seconds 0 --cpu-seconds 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d["host_buffer_path"], indent=1))' | tee gpurun_out/r03f_hostpath.txt
