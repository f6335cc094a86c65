cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_default_path.py tests/test_hip_dist.py tests/test_hip_swinir.py -q 2>&1 | tail -8 > gpurun_out/r04_g3_pytest.log
timeout 900 python tools/ddp_contention.py > gpurun_out/r04_g3_ddp_contention.jsonl 2> gpurun_out/r04_g3_ddp_contention.err
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then export NEOSR_AMD_LIB=$PWD/experiments/old/libneosr_amd.so; else unset NEOSR_AMD_LIB; fi
    python bench.py --config bench_swinir_medium --no-roofline --cpu-budget 0 --no-other-configs --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v swinir', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g3_ab.log
  done
done
unset NEOSR_AMD_LIB
python bench.py --no-roofline --cpu-budget 0 --no-other-configs --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new esrgan', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g3_ab.log
timeout 300 python tools/bench_gemm.py > gpurun_out/r04_g3_gemm_new.log 2>&1
NEOSR_AMD_LIB=$PWD/experiments/old/libneosr_amd.so timeout 300 python tools/bench_gemm.py > gpurun_out/r04_g3_gemm_old.log 2>&1
