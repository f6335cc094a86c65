# final numbers of the round: python bench.py --config <name> for the five BASELINE configs (default flags: roofline pass and
# CPU baseline included), JSON lines -> gpurun_out/final_${1:-r04}_<config>.json
cd /root/repo
mkdir -p gpurun_out
TAG=${1:-r05}
for c in bench_esrgan bench_compact bench_esrgan_otf_gan bench_swinir_medium bench_hat_l_otf_gan; do
  python bench.py --config $c --no-other-configs 2> gpurun_out/final_${TAG}_$c.err | tail -1 > gpurun_out/final_${TAG}_$c.json
done
python bench.py 2>/dev/null | tail -1 > gpurun_out/final_${TAG}_default.json
for c in bench_esrgan bench_compact bench_esrgan_otf_gan bench_swinir_medium bench_hat_l_otf_gan; do
  echo "== $c" >> gpurun_out/final_${TAG}_host_overhead.log
  timeout 300 python tools/host_overhead.py $c 2>&1 | tail -1 >> gpurun_out/final_${TAG}_host_overhead.log
done
