# A/B of the chain kernel change (abort word) + the full GPU suite
cd /root/repo
mkdir -p gpurun_out
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then export NEOSR_AMD_LIB=$PWD/experiments/old/libneosr_amd.so; else unset NEOSR_AMD_LIB; fi
    python bench.py --no-roofline --cpu-budget 0 --no-other-configs --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g2_ab.log
  done
done
unset NEOSR_AMD_LIB
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r04_g2_pytest.log
timeout 300 python tools/hat_l_sens.py > gpurun_out/r04_g2_hat_l_sens.log 2>&1
