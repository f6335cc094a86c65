cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_hat.py tests/test_hip_blocks.py tests/test_hip_cfgs.py -q 2>&1 | tail -8 > gpurun_out/r04_g15_pytest.log
timeout 300 python tools/bench_fattn.py > gpurun_out/r04_g15_fattn.log 2>&1
for rep in 1 2; do
python bench.py --config bench_hat_l_otf_gan --no-roofline --cpu-budget 0 --no-other-configs --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hat_l', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g15_ab.log
done
