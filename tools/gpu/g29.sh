cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_direct_grads.py -q -x 2>&1 | tail -15 > gpurun_out/r04_g29.log
for i in 1 2 3; do python bench.py --config bench_compact --steps 300 --warmup 20 --no-roofline --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done >> gpurun_out/r04_g29.log 2>&1
NEOSR_AMD_DIRECT_GRADS=0 python bench.py --config bench_compact --steps 300 --warmup 20 --no-roofline --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('off', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g29.log 2>&1
python tools/host_bwd_compact.py 2>&1 | tail -1 >> gpurun_out/r04_g29.log
python tools/host_profile.py bench_compact 300 > gpurun_out/r04_g29_hostprof_compact.log 2>&1
