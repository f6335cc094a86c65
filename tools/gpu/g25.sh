cd /root/repo
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
  timeout 900 python -m pytest tests/test_hip_chain.py tests/test_hip_default_path.py tests/test_hip_blocks.py tests/test_hip_hat.py tests/test_hip_dist.py -q -x 2>&1 | tail -1 >> gpurun_out/r04_g25_repeat.log
done
timeout 1500 python -m pytest tests -m gpu -q -x -p no:randomly 2>&1 | tail -2 >> gpurun_out/r04_g25_repeat.log
