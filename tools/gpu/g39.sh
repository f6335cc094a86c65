cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -2 > gpurun_out/r04_g39.log
for i in 1 2 3; do timeout 600 python -m pytest tests/test_hip_blocks.py tests/test_hip_hat.py tests/test_hip_cfgs.py -q -x 2>&1 | tail -1 >> gpurun_out/r04_g39.log; done
python bench.py --config bench_hat_l_otf_gan 2> gpurun_out/final_r04d_bench_hat_l_otf_gan.err | tail -1 > gpurun_out/final_r04d_bench_hat_l_otf_gan.json
python tools/host_overhead.py bench_hat_l_otf_gan 2>&1 | tail -1 >> gpurun_out/r04_g39.log
