cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r04_g22_pytest_all.log
for c in bench_swinir_medium bench_hat_l_otf_gan; do
  bash tools/profile_cfg.sh r04 $c > gpurun_out/r04_g22_prof_$c.log 2>&1
  bash tools/profile_sq.sh r04 $c >> gpurun_out/r04_g22_prof_$c.log 2>&1
done
bash tools/gpu/final.sh r04b
