cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_blocks.py tests/test_hip_swinir.py tests/test_hip_cfgs.py -q -x 2>&1 | tail -15 > gpurun_out/r04_g4_pytest.log
for v in 0 1; do
  echo "== NEOSR_AMD_BLOCK_PLANS=$v" >> gpurun_out/r04_g4_host.log
  NEOSR_AMD_BLOCK_PLANS=$v timeout 300 python tools/host_overhead.py bench_swinir_medium 2>&1 | tail -1 >> gpurun_out/r04_g4_host.log
  NEOSR_AMD_BLOCK_PLANS=$v python bench.py --config bench_swinir_medium --no-roofline --cpu-budget 0 --no-other-configs --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plans=$v swinir', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g4_host.log
done
