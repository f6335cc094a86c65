cd /root/repo
mkdir -p gpurun_out
timeout 600 python tools/host_ops.py bench_hat_l_otf_gan > gpurun_out/r04_g8_host_ops_hat_l.log 2>&1
timeout 600 python tools/host_ops.py bench_swinir_medium > gpurun_out/r04_g8_host_ops_swinir.log 2>&1
timeout 600 python tools/host_ops.py bench_esrgan_otf_gan > gpurun_out/r04_g8_host_ops_cfg2.log 2>&1
