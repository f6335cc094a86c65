cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_blocks.py -q 2>&1 | tail -4 > gpurun_out/r04_g23_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_g23_smoke.log 2>&1
python bench.py --steps 400 --warmup 5 --no-roofline --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 > gpurun_out/r04_g23_soak.json
python -c "
from neosr_amd import _C
print('chain status', _C.load().neosr_conv_chain_status())" >> gpurun_out/r04_g23_smoke.log 2>&1
