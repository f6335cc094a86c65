cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_blocks.py tests/test_hip_hat.py -q 2>&1 | tail -8 > gpurun_out/r04_g7_pytest.log
for rep in 1 2; do
for s in 1 2; do
  NEOSR_AMD_BLOCK_STREAMS=$s python bench.py --config bench_hat_l_otf_gan --no-roofline --cpu-budget 0 --no-other-configs --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hat_l streams=$s', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g7_ab.log
  NEOSR_AMD_BLOCK_STREAMS=$s python bench.py --config bench_swinir_medium --no-roofline --cpu-budget 0 --no-other-configs --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('swinir streams=$s', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g7_ab.log
done
for wb in 0 1; do
  NEOSR_AMD_CHAIN_WB=$wb python bench.py --no-roofline --cpu-budget 0 --no-other-configs --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('esrgan chain_wb=$wb', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g7_ab.log
done
done
NEOSR_AMD_CHAIN_WB=1 timeout 600 python -m pytest tests/test_hip_chain.py tests/test_hip_default_path.py -q 2>&1 | tail -5 > gpurun_out/r04_g7_pytest_wb.log
hipcc --offload-arch=gfx950 -O3 tools/micro/ub.hip -o /tmp/ub 2>/dev/null && /tmp/ub > gpurun_out/r04_g7_ub.log 2>&1
timeout 200 python tools/micro/gemm_probe.py > gpurun_out/r04_g7_gemm_probe.log 2>&1
