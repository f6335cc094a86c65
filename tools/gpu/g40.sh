cd /root/repo
mkdir -p gpurun_out
bash tools/profile_cfg.sh r04 bench_hat_l_otf_gan > gpurun_out/r04_g40_prof.log 2>&1
bash tools/profile_sq.sh r04 bench_hat_l_otf_gan >> gpurun_out/r04_g40_prof.log 2>&1
