cd /root/repo
mkdir -p gpurun_out
env | grep -i "HIP_\|HSA_\|GPU_\|ROC" > gpurun_out/r04_g9_env.log
timeout 300 python tools/host_block.py 4 > gpurun_out/r04_g9_host_block.log 2>&1
HIP_FORCE_DEV_KERNARG=1 timeout 300 python tools/host_block.py 4 >> gpurun_out/r04_g9_host_block.log 2>&1
for k in 0 1; do
for c in bench_esrgan bench_swinir_medium bench_hat_l_otf_gan bench_compact; do
  HIP_FORCE_DEV_KERNARG=$k python bench.py --config $c --no-roofline --cpu-budget 0 --no-other-configs --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c devkernarg=$k', d['value'], d['ms_per_step'])" >> gpurun_out/r04_g9_ab.log
done
done
