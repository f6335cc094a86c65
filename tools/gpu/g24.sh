cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_gan.py tests/test_hip_cfgs.py tests/test_hip_optim.py tests/test_hip_adan.py tests/test_hip_fsam.py tests/test_hip_ckpt.py tests/test_hip_val.py -q 2>&1 | tail -5 > gpurun_out/r04_g24_pytest.log
for rep in 1 2; do
for v in 0 1; do
for c in bench_esrgan_otf_gan bench_hat_l_otf_gan; do
  NEOSR_AMD_ARENA_EPOCH=$v python bench.py --config $c --no-roofline --cpu-budget 0 --steps 15 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c arena_epoch=$v', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g24_ab.log
done
done
done
