mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/r03g_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03g_pytest.log; tail -6 gpurun_out/r03g_pytest.log
bash tools/evidence.sh r03 2>&1 | tail -5
B="timeout -k 10 400 python bench.py"
$B --config c2 --no-cpu > gpurun_out/r03_c2_bench.json 2>/dev/null
$B --config c4shard --no-cpu > gpurun_out/r03_c4shard_bench.json 2>/dev/null
$B --config ref128x --cpu-seconds 6 > gpurun_out/r03_ref128x_bench.json 2>/dev/null
PVW_MAC_PACKED=0 $B --tuning-library --config ref128x --no-cpu --no-probe --sustain-seconds 0 > gpurun_out/r03_ref128x_unpacked_bench.json 2>/dev/null
$B --path decrypt --config c5shard > gpurun_out/r03_decrypt_c5shard_bench.json 2>/dev/null
$B --path decrypt --config c5full --steps 10 --warmup 2 > gpurun_out/r03_decrypt_c5full_bench.json 2>/dev/null
$B --dealers 64 --no-cpu > gpurun_out/r03_multi64_bench.json 2>/dev/null
$B --path keygen --no-cpu > gpurun_out/r03_keygen_bench.json 2>/dev/null
for f in gpurun_out/r03_*_bench.json; do python - "$f" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print(sys.argv[1].split("/")[-1], round(d["value"]), round(d["ms_per_step"] * 1000, 1), "us/step", r.get("kernel"), round(r.get("avg_launch_us", 0), 1), "frac", round(r.get("frac", 0), 3), "streamed", round(r.get("frac_of_streamed_bytes", 0) or 0, 3))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
