cd /root/repo
mkdir -p gpurun_out
echo "== hat_l patch 64 (as named)" > gpurun_out/r04_g19_host.log
timeout 300 python tools/host_overhead.py bench_hat_l_otf_gan 2>&1 | tail -1 >> gpurun_out/r04_g19_host.log
echo "== hat_l patch 32 (same launches, ~1/4 of the device work)" >> gpurun_out/r04_g19_host.log
timeout 300 python tools/host_overhead.py bench_hat_l_otf_gan 32 2>&1 | tail -1 >> gpurun_out/r04_g19_host.log
echo "== hat_l patch 16" >> gpurun_out/r04_g19_host.log
timeout 300 python tools/host_overhead.py bench_hat_l_otf_gan 16 2>&1 | tail -1 >> gpurun_out/r04_g19_host.log
echo "== hat_l patch 32, NEOSR_AMD_BLOCK_PLANS=0" >> gpurun_out/r04_g19_host.log
NEOSR_AMD_BLOCK_PLANS=0 timeout 300 python tools/host_overhead.py bench_hat_l_otf_gan 32 2>&1 | tail -1 >> gpurun_out/r04_g19_host.log
echo "== esrgan_otf_gan patch 32" >> gpurun_out/r04_g19_host.log
timeout 300 python tools/host_overhead.py bench_esrgan_otf_gan 32 2>&1 | tail -1 >> gpurun_out/r04_g19_host.log
