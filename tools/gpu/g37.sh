cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_blocks.py -q -x 2>&1 | tail -5 > gpurun_out/r04_g37.log
for m in 1 3 1 3; do
  NEOSR_AMD_BLOCK_STREAMS=$m python bench.py --config bench_hat_l_otf_gan --steps 15 --warmup 3 --no-roofline --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams=$m hat_l', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g37.log
done
bash tools/trace_overlap.sh bench_hat_l_otf_gan NEOSR_AMD_BLOCK_STREAMS=3 >> gpurun_out/r04_g37.log 2>&1
