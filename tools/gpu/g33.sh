cd /root/repo
mkdir -p gpurun_out
bash tools/trace_by_grid.sh bench_hat_l_otf_gan 3
bash tools/trace_by_grid.sh bench_swinir_medium 3
