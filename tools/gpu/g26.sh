cd /root/repo
mkdir -p gpurun_out
python tools/host_profile.py bench_compact 300 > gpurun_out/r04_g26_hostprof_compact.log 2>&1
python tools/host_profile.py bench_esrgan 30 > gpurun_out/r04_g26_hostprof_esrgan.log 2>&1
