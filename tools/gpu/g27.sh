cd /root/repo
mkdir -p gpurun_out
for i in 1 2 3; do python bench.py --config bench_compact --steps 300 --warmup 20 --no-roofline --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done > gpurun_out/r04_g27_compact.log 2>&1
python tools/host_profile.py bench_compact 300 > gpurun_out/r04_g27_hostprof_compact.log 2>&1
python tools/host_overhead.py bench_compact >> gpurun_out/r04_g27_compact.log 2>&1
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_default_path.py tests/test_hip_models.py -q -x 2>&1 | tail -2 >> gpurun_out/r04_g27_compact.log
