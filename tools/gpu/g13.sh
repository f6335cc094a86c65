cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_hip_blocks.py tests/test_hip_hat.py -q 2>&1 | tail -5 > gpurun_out/r04_g13_pytest.log
for rep in 1 2; do
  python bench.py --config bench_compact --no-roofline --cpu-budget 0 --no-other-configs --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('compact', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g13_ab.log
done
for c in bench_esrgan bench_hat_l_otf_gan; do
  python bench.py --config $c --no-roofline --cpu-budget 0 --no-other-configs --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g13_ab.log
done
timeout 300 python tools/host_overhead.py bench_compact 2>&1 | tail -1 >> gpurun_out/r04_g13_ab.log
timeout 300 python tools/host_block.py 4 2>&1 | tail -2 >> gpurun_out/r04_g13_ab.log
