cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_hip_adan.py -q -k "compact or Compact or adan or trajectory" 2>&1 | tail -15 > gpurun_out/r04_g11_pytest_compact.log
for rep in 1 2; do
for v in 0 1; do
  NEOSR_AMD_COMPACT_W4=$v python bench.py --config bench_compact --no-roofline --cpu-budget 0 --no-other-configs --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('compact w4=$v', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g11_ab.log
done
done
for c in bench_esrgan bench_esrgan_otf_gan bench_swinir_medium bench_hat_l_otf_gan; do
  python bench.py --config $c --no-roofline --cpu-budget 0 --no-other-configs --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g11_ab.log
done
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r04_g11_pytest_all.log
