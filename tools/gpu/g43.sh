cd /root/repo
mkdir -p gpurun_out
L=gpurun_out/r04_g43.log
: > $L
for i in 1 2 3 4 5 6 7 8; do timeout 600 python -m pytest tests/test_hip_blocks.py tests/test_hip_hat.py tests/test_hip_direct_grads.py -q -x 2>&1 | tail -1 >> $L; done
python bench.py --config bench_hat_l_otf_gan --steps 150 --warmup 5 --no-roofline --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hat_l soak', d['value'], d['ms_per_step'], d['final_loss'])" >> $L
python bench.py --steps 1500 --warmup 5 --no-roofline --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('esrgan soak', d['value'], d['ms_per_step'], d['final_loss'])" >> $L
python bench.py --config bench_compact --steps 5000 --warmup 5 --no-roofline --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('compact soak', d['value'], d['ms_per_step'], d['final_loss'])" >> $L
python -c "
from neosr_amd import _C
print('chain status', _C.load().neosr_conv_chain_status())" >> $L 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> $L
