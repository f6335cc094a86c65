cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_swinir.py tests/test_hip_blocks.py tests/test_hip_hat.py -q -x 2>&1 | tail -3 > gpurun_out/r04_g31.log
for lib in new old new old; do
  if [ $lib = old ]; then export NEOSR_AMD_LIB=/root/repo/experiments/old/libneosr_amd.so; else unset NEOSR_AMD_LIB; fi
  python bench.py --config bench_swinir_medium --steps 30 --warmup 5 --no-roofline --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib swinir', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g31.log
done
for lib in new old; do
  if [ $lib = old ]; then export NEOSR_AMD_LIB=/root/repo/experiments/old/libneosr_amd.so; else unset NEOSR_AMD_LIB; fi
  python bench.py --config bench_hat_l_otf_gan --steps 15 --warmup 3 --no-roofline --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib hat_l', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g31.log
done
