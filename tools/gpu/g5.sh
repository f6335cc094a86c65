cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_blocks.py -q 2>&1 | tail -30 > gpurun_out/r04_g5_pytest_blocks.log
timeout 1200 python -m pytest tests/test_hip_swinir.py tests/test_hip_cfgs.py tests/test_hip_hat.py tests/test_hip_dist.py -q 2>&1 | tail -15 > gpurun_out/r04_g5_pytest.log
for c in bench_swinir_medium bench_hat_l_otf_gan; do
for v in 0 1; do
  echo "== $c NEOSR_AMD_BLOCK_PLANS=$v" >> gpurun_out/r04_g5_host.log
  NEOSR_AMD_BLOCK_PLANS=$v timeout 300 python tools/host_overhead.py $c 2>&1 | tail -1 >> gpurun_out/r04_g5_host.log
  NEOSR_AMD_BLOCK_PLANS=$v python bench.py --config $c --no-roofline --cpu-budget 0 --no-other-configs --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plans=$v', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g5_host.log
done
done
