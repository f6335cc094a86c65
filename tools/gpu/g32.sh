cd /root/repo
mkdir -p gpurun_out
bash tools/gpu/final.sh r04c
bash tools/profile_cfg.sh r04 bench_compact > gpurun_out/r04_g32_prof.log 2>&1
bash tools/profile_sq.sh r04 bench_compact >> gpurun_out/r04_g32_prof.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -2 > gpurun_out/r04_g32_pytest.log
