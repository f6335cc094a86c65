cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_direct_grads.py -q -x 2>&1 | tail -15 > gpurun_out/r04_g30.log
