cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_gan.py tests/test_hip_cfgs.py tests/test_hip_dist.py -q 2>&1 | tail -8 > gpurun_out/r04_g16_pytest.log
for rep in 1 2; do
for v in 0 1; do
  NEOSR_AMD_S2D_WINO=$v python bench.py --config bench_esrgan_otf_gan --no-roofline --cpu-budget 0 --no-other-configs --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 s2d_wino=$v', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g16_ab.log
  NEOSR_AMD_S2D_WINO=$v python bench.py --config bench_hat_l_otf_gan --no-roofline --cpu-budget 0 --no-other-configs --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hat_l s2d_wino=$v', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g16_ab.log
done
done
