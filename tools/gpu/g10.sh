cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_swinir.py tests/test_hip_blocks.py -q 2>&1 | tail -8 > gpurun_out/r04_g10_pytest.log
for f in 0 1; do
  echo "== NEOSR_GEMM_FULLK=$f B=8" >> gpurun_out/r04_g10_gemm.log
  NEOSR_GEMM_FULLK=$f timeout 300 python tools/bench_gemm.py 8 2>&1 | grep "NT" >> gpurun_out/r04_g10_gemm.log
  echo "== NEOSR_GEMM_FULLK=$f B=4" >> gpurun_out/r04_g10_gemm.log
  NEOSR_GEMM_FULLK=$f timeout 300 python tools/bench_gemm.py 4 2>&1 | grep "NT" >> gpurun_out/r04_g10_gemm.log
done
for rep in 1 2; do
for f in 0 1; do
for c in bench_swinir_medium bench_hat_l_otf_gan; do
  NEOSR_GEMM_FULLK=$f python bench.py --config $c --no-roofline --cpu-budget 0 --no-other-configs --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c fullk=$f', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g10_ab.log
done
done
done
