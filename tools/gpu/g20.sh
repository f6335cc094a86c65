cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_blocks.py tests/test_hip_swinir.py tests/test_hip_hat.py tests/test_hip_cfgs.py -q 2>&1 | tail -6 > gpurun_out/r04_g20_pytest.log
for rep in 1 2; do
for v in 0 1; do
for c in bench_swinir_medium bench_hat_l_otf_gan; do
  NEOSR_AMD_TN_GROUP=$v python bench.py --config $c --no-roofline --cpu-budget 0 --no-other-configs --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c tn_group=$v', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g20_ab.log
done
done
done
