cd /root/repo
mkdir -p gpurun_out
python tools/host_bwd_compact.py > gpurun_out/r04_g28.log 2>&1
