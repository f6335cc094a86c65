cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_blocks.py tests/test_hip_hat.py tests/test_hip_cfgs.py -q 2>&1 | tail -30 > gpurun_out/r04_g6_pytest.log
timeout 600 python tools/host_profile.py bench_hat_l_otf_gan > gpurun_out/r04_g6_host_profile_hat_l.log 2>&1
for f in 0 1; do
  NEOSR_AMD_FATTN_FUSED=$f python bench.py --config bench_hat_l_otf_gan --no-roofline --cpu-budget 0 --no-other-configs --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused=$f', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g6_fattn.log
done
NEOSR_AMD_FATTN_FUSED=1 timeout 300 python tools/bench_fattn.py > gpurun_out/r04_g6_bench_fattn.log 2>&1
NEOSR_AMD_FATTN_FUSED=0 timeout 300 python tools/bench_fattn.py >> gpurun_out/r04_g6_bench_fattn.log 2>&1
