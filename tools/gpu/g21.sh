cd /root/repo
mkdir -p gpurun_out
cp neosr_amd/lib/libneosr_amd.so /tmp/new.so
timeout 900 python -m pytest tests/test_hip_swinir.py tests/test_hip_blocks.py tests/test_hip_hat.py tests/test_hip_cfgs.py -q 2>&1 | tail -6 > gpurun_out/r04_g21_pytest.log
for rep in 1 2; do
for v in old new; do
  if [ $v = old ]; then export NEOSR_AMD_LIB=$PWD/experiments/old/libneosr_amd.so; else unset NEOSR_AMD_LIB; fi
for c in bench_swinir_medium bench_hat_l_otf_gan; do
  python bench.py --config $c --no-roofline --cpu-budget 0 --no-other-configs --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $v', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g21_ab.log
done
done
done
