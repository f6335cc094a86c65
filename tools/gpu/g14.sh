cd /root/repo
mkdir -p gpurun_out
timeout 200 python tools/bench_conv_hat.py 4 > gpurun_out/r04_g14_conv_hat.log 2>&1
timeout 200 python tools/bench_conv_hat.py 8 >> gpurun_out/r04_g14_conv_hat.log 2>&1
bash tools/profile_cfg.sh r04 bench_esrgan > gpurun_out/r04_g14_prof_esrgan.log 2>&1
bash tools/profile_sq.sh r04 bench_esrgan >> gpurun_out/r04_g14_prof_esrgan.log 2>&1
bash tools/profile_cfg.sh r04 bench_compact > gpurun_out/r04_g14_prof_compact.log 2>&1
bash tools/profile_sq.sh r04 bench_compact >> gpurun_out/r04_g14_prof_compact.log 2>&1
