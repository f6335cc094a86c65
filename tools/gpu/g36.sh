cd /root/repo
mkdir -p gpurun_out
bash tools/trace_overlap.sh bench_hat_l_otf_gan NEOSR_AMD_BLOCK_STREAMS=3 > gpurun_out/r04_g36.log 2>&1
bash tools/trace_overlap.sh bench_hat_l_otf_gan NEOSR_AMD_BLOCK_STREAMS=1 >> gpurun_out/r04_g36.log 2>&1
