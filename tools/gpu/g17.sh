cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_swinir.py tests/test_hip_blocks.py tests/test_hip_hat.py -q 2>&1 | tail -5 > gpurun_out/r04_g17_pytest.log
timeout 300 python tools/bench_gemm.py 8 2>&1 | grep "NT" > gpurun_out/r04_g17_gemm.log
timeout 300 python tools/bench_gemm.py 4 2>&1 | grep "NT" >> gpurun_out/r04_g17_gemm.log
for rep in 1 2; do
for c in bench_swinir_medium bench_hat_l_otf_gan; do
  python bench.py --config $c --no-roofline --cpu-budget 0 --no-other-configs --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g17_ab.log
done
done
