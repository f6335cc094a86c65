set -x
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_g1_pytest.log
python bench.py > gpurun_out/r04_g1_bench_esrgan.json 2> gpurun_out/r04_g1_bench_esrgan.err
timeout 300 python tools/memcpy_profile.py bench_esrgan > gpurun_out/r04_g1_memcpy_esrgan.log 2>&1
timeout 300 python tools/hat_l_sens.py > gpurun_out/r04_g1_hat_l_sens.log 2>&1
for c in bench_esrgan bench_compact bench_esrgan_otf_gan bench_swinir_medium bench_hat_l_otf_gan; do
  echo "== $c" >> gpurun_out/r04_g1_host_overhead.log
  timeout 300 python tools/host_overhead.py $c >> gpurun_out/r04_g1_host_overhead.log 2>&1
done
timeout 300 python tools/memcpy_profile.py bench_hat_l_otf_gan > gpurun_out/r04_g1_memcpy_hat_l.log 2>&1
