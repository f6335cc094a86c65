cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -q -k "compact or Compact" 2>&1 | tail -5 > gpurun_out/r04_g12_pytest.log
for rep in 1 2; do
for v in 0 1; do
  NEOSR_AMD_COMPACT_W4=$v python bench.py --config bench_compact --no-roofline --cpu-budget 0 --no-other-configs --steps 200 --warmup 20 2>gpurun_out/r04_g12_err_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('compact w4=$v', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g12_ab.log
done
done
NEOSR_AMD_COMPACT_W4=1 python bench.py --config bench_compact --cpu-budget 0 --no-other-configs --steps 50 --warmup 5 > gpurun_out/r04_g12_compact_roof.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_g12 -o trace --output-format csv -- python /root/repo/bench.py --config bench_compact --cpu-budget 0 --no-other-configs --no-roofline --steps 50 --warmup 5 > /root/repo/gpurun_out/r04_g12_trace.log 2>&1
cp $(find /root/repo/gpurun_out/prof_g12 -name '*kernel_stats.csv' | head -1) /root/repo/gpurun_out/r04_g12_compact_kernel_stats.csv
rm -rf /root/repo/gpurun_out/prof_g12
