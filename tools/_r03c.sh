mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "mac_rows or packed or prepare or switches or seed_mode or config3 or ragged or rccl or forced or decode or decrypt or config5 or golden" > gpurun_out/r03c_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03c_pytest.log; tail -5 gpurun_out/r03c_pytest.log
J='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], round(d["value"]), round(d["ms_per_step"]*1000,1), round(r["avg_launch_us"],1), r.get("kernel"), {k: round(v*1000,1) for k,v in d["kernel_ms_per_step"].items() if v})'
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --no-cpu --no-probe --sustain-seconds 0 2>/dev/null | python -c "$J" fused
  PVW_MAC_FRONT=0 timeout -k 10 200 python bench.py --tuning-library --no-cpu --no-probe --sustain-seconds 0 2>/dev/null | python -c "$J" classic
  PVW_FRONT_GUARD_ALL=1 timeout -k 10 200 python bench.py --tuning-library --no-cpu --no-probe --sustain-seconds 0 2>/dev/null | python -c "$J" fused_guard_all
done 2>&1 | tee gpurun_out/r03c_front_ab.txt
for c in c2 c4shard ref128x; do
  timeout -k 10 300 python bench.py --no-cpu --no-probe --sustain-seconds 0 --config $c 2>/dev/null | python -c "$J" $c
  PVW_MAC_FRONT=0 timeout -k 10 300 python bench.py --tuning-library --no-cpu --no-probe --sustain-seconds 0 --config $c 2>/dev/null | python -c "$J" ${c}_classic
done 2>&1 | tee gpurun_out/r03c_configs.txt
PVW_MAC_PACKED=0 timeout -k 10 300 python bench.py --tuning-library --no-cpu --no-probe --sustain-seconds 0 --config ref128x 2>/dev/null | python -c "$J" ref128x_unpacked | tee -a gpurun_out/r03c_configs.txt
for ch in 0 768 512 256; do
  PVW_DECRYPT_CHUNK=$ch timeout -k 10 300 python bench.py --tuning-library --path decrypt --config c5shard --steps 30 --warmup 5 2>/dev/null | python -c "$J" c5shard_chunk$ch
done 2>&1 | tee gpurun_out/r03c_c5shard.txt
timeout -k 10 300 python bench.py --path decrypt --config c5full --steps 10 --warmup 2 2>/dev/null | python -c "$J" c5full | tee -a gpurun_out/r03c_c5shard.txt
for v in 2 8; do
  PVW_FINISH_VPB=$v timeout -k 10 300 python bench.py --tuning-library --dealers 64 --no-cpu --no-probe --sustain-seconds 0 2>/dev/null | python -c "$J" multi64_vpb$v
  PVW_FINISH_VPB=$v timeout -k 10 300 python bench.py --tuning-library --path keygen --no-cpu 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],3))' keygen_vpb$v
done 2>&1 | tee gpurun_out/r03c_finish.txt
